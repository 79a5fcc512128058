"""mgx — MI355X-native geometric multigrid for the 2-D Poisson problem.

Drop-in for the hot path of nikhilTkur/Multigrid_Nikhil_C- (smoother, residual,
restriction, prolongation, V-cycle / FMG schedules).  The product
is libmgx.so (C-ABI: include/mgx.h) built from csrc/; this package is the thin
ctypes view of it plus the multi-GPU slab driver.  The directory name carries
the reference's name, so import it through __graft_entry__.load_package()
(registered as `multigrid_nikhil_c_amd`).
"""
from .binding import (  # noqa: F401
    ARITH_FMA, ARITH_SEPARATE, BOTTOM_EXACT, BOTTOM_SMOOTH, DTYPE_F32, DTYPE_F64, DTYPE_MIXED, EXPORTS, LIB_PATH,
    OPERATOR_POISSON, OPERATOR_STENCIL5, RESTRICT_CONSISTENT, RESTRICT_FW16, RESTRICT_INJECT, RESTRICT_INJECT4, SCHEDULE_FMG, SCHEDULE_V, SMOOTHER_JACOBI,
    SMOOTHER_RBGS, VEC_B, VEC_R, VEC_U, Config, DistLevel, DistOp, MgxError, Multigrid, Plan, Slab, Transport, Xfer,
    default_config, lib, rccl_unique_id, runtime_libs,
)
