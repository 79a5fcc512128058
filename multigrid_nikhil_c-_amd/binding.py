"""ctypes binding of libmgx.so (include/mgx.h).

The method names follow the reference's free functions so that callers read
like the reference's own driver (PS = Poissons_SYCL.cpp):
    jacobirelaxation   PS:125      restriction2d     PS:531
    interpolation2d    PS:337      vcyclemultigrid   PS:575
    fullmultigrid      PS:629      solve             PS:727 (main's call)

There is no CPU fallback: if libmgx.so is missing or no HIP device is usable
every entry point raises.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# MGX_LIBMGX_PATH: another build of the same library (A/B measurements inside one GPU call)
LIB_PATH = os.environ.get("MGX_LIBMGX_PATH") or os.path.join(_HERE, "libmgx.so")

SMOOTHER_JACOBI, SMOOTHER_RBGS = 0, 1
DTYPE_F32, DTYPE_F64, DTYPE_MIXED = 0, 1, 2
SCHEDULE_V, SCHEDULE_FMG = 0, 1
RESTRICT_CONSISTENT, RESTRICT_FW16, RESTRICT_INJECT, RESTRICT_INJECT4 = 0, 1, 2, 3
OPERATOR_POISSON, OPERATOR_STENCIL5 = 0, 1
BOTTOM_EXACT, BOTTOM_SMOOTH = 0, 1
ARITH_SEPARATE, ARITH_FMA = 0, 1
VEC_U, VEC_B, VEC_R = 0, 1, 2
PROF_SMOOTH_FINE, PROF_RESTRICT_FINE, PROF_PROLONG_FINE, PROF_NORM_FINE, PROF_COARSE, PROF_COUNT = 0, 1, 2, 3, 4, 5

# every symbol include/mgx.h declares (tests check the library exports them all)
EXPORTS = [
    "mgx_config_default", "mgx_create", "mgx_destroy", "mgx_last_error", "mgx_status_string",
    "mgx_level_n", "mgx_set_rhs", "mgx_set_rhs_dirichlet", "mgx_set_guess", "mgx_get_solution", "mgx_set_level",
    "mgx_get_level", "mgx_set_level_device", "mgx_get_level_device", "mgx_zero_level", "mgx_fill_rhs", "mgx_fill_guess_random", "mgx_smooth", "mgx_residual",
    "mgx_restrict", "mgx_restrict_rhs", "mgx_prolong_add", "mgx_prolong", "mgx_bottom_solve",
    "mgx_residual_norm", "mgx_vcycle", "mgx_vcycle_zero", "mgx_fmg", "mgx_solve", "mgx_profile_reset",
    "mgx_profile_get", "mgx_time_smoother", "mgx_synchronize", "mgx_graphs_cached", "mgx_slab_cycle", "mgx_level_pitch",
    "mgx_slab_jacobi", "mgx_slab_rbgs", "mgx_slab_restrict", "mgx_slab_prolong",
    "mgx_slab_residual_sumsq", "mgx_slab_scratch_doubles",
    "mgx_plan_create", "mgx_plan_destroy", "mgx_plan_last_error", "mgx_plan_cut_level", "mgx_plan_level",
    "mgx_plan_cut_share", "mgx_plan_guess_set", "mgx_plan_vcycle", "mgx_plan_norm", "mgx_plan_fmg", "mgx_rccl_unique_id",
    "mgx_create_rank", "mgx_dist_exchanges", "mgx_dist_overlapped", "mgx_memcpy_d2h", "mgx_memcpy_h2d", "mgx_runtime_libs",
    "mgx_set_stencil", "mgx_set_coefficient", "mgx_get_stencil",
]
MAX_GPUS = 16
(DOP_EXCHANGE, DOP_ZERO_U, DOP_CYCLE, DOP_SMOOTH, DOP_RESTRICT, DOP_PROLONG, DOP_GATHER_CUT, DOP_COARSE, DOP_SUMSQ,
 DOP_ALLREDUCE_NORM, DOP_RESTRICT_RHS, DOP_PROLONG_SET, DOP_COARSE_FMG) = range(1, 14)


class Config(C.Structure):
    _fields_ = [
        ("finest_level", C.c_int), ("coarsest_level", C.c_int),
        ("mu0", C.c_int), ("mu1", C.c_int), ("mu2", C.c_int),
        ("omega", C.c_double),
        ("smoother", C.c_int), ("dtype", C.c_int), ("schedule", C.c_int),
        ("restrict_mode", C.c_int), ("bottom", C.c_int),
        ("device", C.c_int), ("profile", C.c_int),
        ("n_gpus", C.c_int), ("cut_level", C.c_int), ("devices", C.c_int * MAX_GPUS),
        ("arith", C.c_int), ("op", C.c_int),
    ]


class Stats(C.Structure):
    _fields_ = [
        ("cycles", C.c_int), ("converged", C.c_int),
        ("initial_residual", C.c_double), ("final_residual", C.c_double),
        ("seconds", C.c_double), ("fine_updates", C.c_double), ("history_len", C.c_int),
    ]


class Profile(C.Structure):
    _fields_ = [("ms", C.c_double * PROF_COUNT), ("launches", C.c_longlong * PROF_COUNT),
                ("sweeps", C.c_longlong * PROF_COUNT)]


class Slab(C.Structure):
    _fields_ = [("level", C.c_int), ("dtype", C.c_int), ("rows", C.c_int), ("row0", C.c_int), ("arith", C.c_int)]


class DistOp(C.Structure):
    """mgx_dist_op: one operation of a slab plan"""
    _fields_ = [("op", C.c_int), ("level", C.c_int), ("which", C.c_int), ("depth", C.c_int),
                ("row_lo", C.c_int), ("row_hi", C.c_int), ("mu", C.c_int), ("pre", C.c_int), ("post", C.c_int),
                ("crow_lo", C.c_int), ("crow_hi", C.c_int), ("coarse_is_cut", C.c_int)]


class DistLevel(C.Structure):
    _fields_ = [("level", C.c_int), ("N", C.c_int), ("own_lo", C.c_int), ("own_hi", C.c_int), ("halo", C.c_int),
                ("row0", C.c_int), ("rows", C.c_int)]


class Xfer(C.Structure):
    _fields_ = [("send", C.c_int), ("peer", C.c_int), ("ptr", C.c_void_p), ("bytes", C.c_size_t)]


SENDRECV_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.POINTER(Xfer), C.c_void_p)
ALLGATHER_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p)
ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_double))


class Transport(C.Structure):
    """mgx_transport: what moves between the slabs of different processes"""
    _fields_ = [("ctx", C.c_void_p), ("sendrecv", SENDRECV_FN), ("allgather", ALLGATHER_FN), ("allreduce_sum", ALLREDUCE_FN)]


class MgxError(RuntimeError):
    pass


def rccl_unique_id() -> bytes:
    """128 bytes naming a new RCCL communicator (rank 0 calls it, the launcher hands them to every rank)"""
    buf = C.create_string_buffer(128)
    if lib().mgx_rccl_unique_id(buf) != 0:
        raise MgxError("mgx_rccl_unique_id failed")
    return buf.raw


_lib = None


def lib() -> C.CDLL:
    """Load libmgx.so; fails loudly when the HIP extension has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise MgxError(
            f"{LIB_PATH} not found: build it with `make -C multigrid_nikhil_c-_amd/csrc` "
            "(or __graft_entry__.build()); there is no CPU fallback")
    L = C.CDLL(LIB_PATH)
    vp, ip, dp = C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_double)
    L.mgx_config_default.argtypes = [C.POINTER(Config)]
    L.mgx_create.argtypes = [C.POINTER(Config), C.POINTER(vp)]
    L.mgx_destroy.argtypes = [vp]
    L.mgx_last_error.argtypes = [vp]
    L.mgx_last_error.restype = C.c_char_p
    L.mgx_status_string.argtypes = [C.c_int]
    L.mgx_status_string.restype = C.c_char_p
    L.mgx_level_n.argtypes = [C.c_int]
    L.mgx_level_pitch.argtypes = [C.c_int, C.c_int]
    L.mgx_level_pitch.restype = C.c_long
    for name in ("mgx_set_rhs", "mgx_set_guess", "mgx_get_solution"):
        getattr(L, name).argtypes = [vp, vp, C.c_size_t]
    L.mgx_set_rhs_dirichlet.argtypes = [vp, vp, C.c_size_t, vp, C.c_size_t]
    L.mgx_set_level.argtypes = [vp, C.c_int, C.c_int, vp, C.c_size_t]
    L.mgx_get_level.argtypes = [vp, C.c_int, C.c_int, vp, C.c_size_t]
    L.mgx_set_level_device.argtypes = [vp, C.c_int, C.c_int, vp]
    L.mgx_get_level_device.argtypes = [vp, C.c_int, C.c_int, vp]
    L.mgx_zero_level.argtypes = [vp, C.c_int, C.c_int]
    L.mgx_fill_rhs.argtypes = [vp, C.c_int, C.c_double]
    L.mgx_fill_guess_random.argtypes = [vp, C.c_uint64]
    L.mgx_smooth.argtypes = [vp, C.c_int, C.c_int]
    for name in ("mgx_residual", "mgx_restrict", "mgx_restrict_rhs", "mgx_prolong_add", "mgx_prolong", "mgx_vcycle"):
        getattr(L, name).argtypes = [vp, C.c_int]
    for name in ("mgx_bottom_solve", "mgx_fmg", "mgx_vcycle_zero", "mgx_profile_reset", "mgx_synchronize", "mgx_graphs_cached"):
        getattr(L, name).argtypes = [vp]
    L.mgx_residual_norm.argtypes = [vp, C.c_int, dp]
    L.mgx_solve.argtypes = [vp, C.c_double, C.c_int, C.POINTER(Stats), dp, C.c_int]
    L.mgx_profile_get.argtypes = [vp, C.POINTER(Profile)]
    L.mgx_time_smoother.argtypes = [vp, C.c_int, dp]
    sp = C.POINTER(Slab)
    L.mgx_slab_jacobi.argtypes = [sp, vp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_double, C.c_int, ip, vp]
    L.mgx_slab_rbgs.argtypes = [sp, vp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, ip, vp]
    L.mgx_slab_cycle.argtypes = [sp, vp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_double, C.c_int, sp, vp, vp,
                                 C.c_int, C.c_int, C.c_int, C.c_int, vp, vp, ip, vp]
    L.mgx_slab_restrict.argtypes = [sp, vp, vp, sp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, vp]
    L.mgx_slab_prolong.argtypes = [sp, vp, sp, vp, C.c_int, C.c_int, C.c_int, vp]
    L.mgx_slab_residual_sumsq.argtypes = [sp, vp, vp, C.c_int, C.c_int, vp, vp, vp]
    L.mgx_slab_scratch_doubles.argtypes = [sp]
    L.mgx_slab_scratch_doubles.restype = C.c_long
    L.mgx_plan_create.argtypes = [C.POINTER(Config), C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(vp)]
    L.mgx_plan_destroy.argtypes = [vp]
    L.mgx_plan_last_error.restype = C.c_char_p
    L.mgx_plan_cut_level.argtypes = [vp]
    L.mgx_plan_level.argtypes = [vp, C.c_int, C.POINTER(DistLevel)]
    L.mgx_plan_cut_share.argtypes = [vp, ip, ip]
    L.mgx_plan_guess_set.argtypes = [vp, C.c_int]
    L.mgx_plan_vcycle.argtypes = [vp, C.POINTER(DistOp), C.c_int]
    L.mgx_plan_norm.argtypes = [vp, C.POINTER(DistOp), C.c_int]
    L.mgx_plan_fmg.argtypes = [vp, C.POINTER(DistOp), C.c_int]
    L.mgx_rccl_unique_id.argtypes = [vp]
    L.mgx_create_rank.argtypes = [C.POINTER(Config), C.c_int, C.c_int, vp, C.POINTER(Transport), C.POINTER(vp)]
    L.mgx_dist_exchanges.argtypes = [vp]
    L.mgx_dist_exchanges.restype = C.c_long
    L.mgx_dist_overlapped.argtypes = [vp]
    L.mgx_dist_overlapped.restype = C.c_long
    L.mgx_memcpy_d2h.argtypes = [vp, vp, C.c_size_t, vp]
    L.mgx_memcpy_h2d.argtypes = [vp, vp, C.c_size_t, vp]
    L.mgx_runtime_libs.argtypes = [C.c_char_p, C.c_size_t]
    L.mgx_set_stencil.argtypes = [vp, C.c_int, vp, vp, vp, vp, vp, C.c_size_t]
    L.mgx_set_coefficient.argtypes = [vp, vp, C.c_size_t]
    L.mgx_get_stencil.argtypes = [vp, C.c_int, C.c_int, vp, C.c_size_t]
    _lib = L
    return L


def runtime_libs() -> list[str]:
    """paths of the ROCm runtime libraries (and libmgx) mapped into this process (mgx_runtime_libs)"""
    buf = C.create_string_buffer(16384)
    if lib().mgx_runtime_libs(buf, len(buf)) < 0:
        return []
    return [p for p in buf.value.decode().splitlines() if p]


def lib_loaded() -> bool:
    """has libmgx.so been loaded into this process yet?"""
    return _lib is not None


def default_config(**kw) -> Config:
    c = Config()
    lib().mgx_config_default(C.byref(c))
    for k, v in kw.items():
        if not hasattr(c, k):
            raise TypeError(f"unknown mgx_config field {k!r}")
        if k == "devices":
            for i, d in enumerate(v):
                c.devices[i] = int(d)
        else:
            setattr(c, k, v)
    return c


class Plan:
    """mgx_plan_*: the multi-GPU V-cycle of slab `g` of `n_slabs` as a list of operations (host
    logic only: usable without a GPU; the CPU tests execute these plans over numpy and gloo)"""

    def __init__(self, n_slabs, g, fold=True, deep=True, **cfg):
        self.cfg = default_config(**cfg)
        self._h = C.c_void_p()
        st = lib().mgx_plan_create(C.byref(self.cfg), n_slabs, g, 1 if fold else 0, 1 if deep else 0, C.byref(self._h))
        if st != 0:
            raise MgxError(f"mgx_plan_create: {lib().mgx_plan_last_error().decode()}")
        self.cut = lib().mgx_plan_cut_level(self._h)
        r0, rows = C.c_int(), C.c_int()
        lib().mgx_plan_cut_share(self._h, C.byref(r0), C.byref(rows))
        self.c_row0, self.c_rows = r0.value, rows.value

    def level(self, level):
        g = DistLevel()
        if lib().mgx_plan_level(self._h, level, C.byref(g)) != 0:
            raise MgxError("mgx_plan_level: level is not distributed")
        return g

    def guess_set(self, all_rows=True):
        lib().mgx_plan_guess_set(self._h, 1 if all_rows else 0)

    def _emit(self, fn):
        cap = 4096
        buf = (DistOp * cap)()
        n = fn(self._h, buf, cap)
        if n < 0:
            raise MgxError(f"plan longer than {cap} operations")
        return [buf[i] for i in range(n)]

    def fmg(self):
        return self._emit(lib().mgx_plan_fmg)

    def vcycle(self):
        return self._emit(lib().mgx_plan_vcycle)

    def norm(self):
        return self._emit(lib().mgx_plan_norm)

    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            lib().mgx_plan_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Multigrid:
    """One mgx_handle: the grid hierarchy of PS:661-690 plus the operators and
    schedules that act on it.  Vectors cross this boundary as 2-D numpy arrays
    in the reference's interior-only n x n layout (PS:227, 291)."""

    def __init__(self, _rank=None, **cfg):
        self.cfg = default_config(**cfg)
        self._h = C.c_void_p()
        self._keep = None
        if _rank is None:
            st = lib().mgx_create(C.byref(self.cfg), C.byref(self._h))
        else:
            rank, world, rccl_id, transport = _rank
            self._keep = (rccl_id, transport)              # callbacks / id bytes must outlive the handle
            st = lib().mgx_create_rank(C.byref(self.cfg), rank, world, rccl_id,
                                       C.byref(transport) if transport is not None else None, C.byref(self._h))
        if st != 0:
            msg = lib().mgx_last_error(None).decode()
            self._h = C.c_void_p()
            raise MgxError(f"mgx_create: {lib().mgx_status_string(st).decode()}: {msg}")

    @classmethod
    def rank(cls, rank, world, rccl_id=None, transport=None, **cfg):
        """mgx_create_rank: this process owns slab `rank` of `world` (one process per GPU).
        rccl_id: the 128 bytes of rccl_unique_id() from rank 0 (built-in RCCL transport), or
        transport: a Transport of the caller's."""
        idbuf = C.create_string_buffer(bytes(rccl_id), 128) if rccl_id is not None else None
        return cls(_rank=(rank, world, idbuf, transport), **cfg)

    def exchanges(self):
        """halo exchanges a multi-GPU handle has performed"""
        return int(lib().mgx_dist_exchanges(self._h))

    def overlapped(self):
        """... of which ran beside the interior rows of the smoothing pass they feed"""
        return int(lib().mgx_dist_overlapped(self._h))

    # -- plumbing --------------------------------------------------------------
    def _chk(self, st, what):
        if st != 0:
            raise MgxError(f"{what}: {lib().mgx_status_string(st).decode()}: {lib().mgx_last_error(self._h).decode()}")

    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            lib().mgx_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def n(self, level=None):
        return (1 << (self.cfg.finest_level if level is None else level)) - 1

    def level_dtype(self, level, which=VEC_U):
        d = self.cfg.dtype
        if d == DTYPE_F64 or (d == DTYPE_MIXED and level == self.cfg.finest_level):
            return np.float64
        return np.float32

    def set_level(self, level, which, a):
        a = np.ascontiguousarray(a, dtype=self.level_dtype(level, which))
        self._chk(lib().mgx_set_level(self._h, level, which, a.ctypes.data, a.size), "mgx_set_level")

    def get_level(self, level, which):
        n = self.n(level)
        a = np.empty((n, n), dtype=self.level_dtype(level, which))
        self._chk(lib().mgx_get_level(self._h, level, which, a.ctypes.data, a.size), "mgx_get_level")
        return a

    def set_level_device(self, level, which, ptr):
        """device-resident input: `ptr` addresses a whole level grid (rows 0..N x pitch)"""
        self._chk(lib().mgx_set_level_device(self._h, level, which, ptr), "mgx_set_level_device")

    def get_level_device(self, level, which, ptr):
        self._chk(lib().mgx_get_level_device(self._h, level, which, ptr), "mgx_get_level_device")

    def zero_level(self, level, which):
        self._chk(lib().mgx_zero_level(self._h, level, which), "mgx_zero_level")

    def set_rhs(self, b):
        self.set_level(self.cfg.finest_level, VEC_B, b)

    def set_rhs_dirichlet(self, b, g_top, g_bottom, g_left, g_right):
        """right-hand side with boundary values folded in: g_top/g_bottom are rows 0 and N
        (N+1 values each), g_left/g_right columns 0 and N on rows 1..N-1"""
        dt = self.level_dtype(self.cfg.finest_level, VEC_B)
        b = np.ascontiguousarray(b, dtype=dt)
        ring = np.ascontiguousarray(np.concatenate([g_top, g_bottom, g_left, g_right]), dtype=dt)
        self._chk(lib().mgx_set_rhs_dirichlet(self._h, b.ctypes.data, b.size, ring.ctypes.data, ring.size),
                  "mgx_set_rhs_dirichlet")

    def set_guess(self, u):
        self.set_level(self.cfg.finest_level, VEC_U, u)

    def get_solution(self):
        return self.get_level(self.cfg.finest_level, VEC_U)

    def get_solution_into(self, a):
        """finest-level U into the caller's n x n array (a rank handle of a multi-GPU job fills the
        rows it owns and leaves the others untouched)"""
        assert a.flags["C_CONTIGUOUS"] and a.dtype == self.level_dtype(self.cfg.finest_level)
        self._chk(lib().mgx_get_solution(self._h, a.ctypes.data, a.size), "mgx_get_solution")
        return a

    # -- general per-level operators (MF:16-41) ---------------------------------------
    def set_stencil(self, level, c, n, s, w, e):
        """ProblemVar::A_sp_dict[level]: five interior n x n coefficient arrays"""
        dt = self.level_dtype(level)
        a = [np.ascontiguousarray(x, dtype=dt) for x in (c, n, s, w, e)]
        self._chk(lib().mgx_set_stencil(self._h, level, *[x.ctypes.data for x in a], a[0].size), "mgx_set_stencil")

    def set_coefficient(self, a_nodes):
        """every level's operator from the nodal coefficient of -div(a grad u) on the finest grid ((N + 1)^2 doubles)"""
        a = np.ascontiguousarray(a_nodes, dtype=np.float64)
        self._chk(lib().mgx_set_coefficient(self._h, a.ctypes.data, a.size), "mgx_set_coefficient")

    def get_stencil(self, level, which):
        n = self.n(level)
        a = np.empty((n, n), dtype=self.level_dtype(level))
        self._chk(lib().mgx_get_stencil(self._h, level, which, a.ctypes.data, a.size), "mgx_get_stencil")
        return a

    def fill_rhs(self, kind=0, f=4.0):
        self._chk(lib().mgx_fill_rhs(self._h, kind, f), "mgx_fill_rhs")

    def fill_guess_random(self, seed=12345):
        self._chk(lib().mgx_fill_guess_random(self._h, seed), "mgx_fill_guess_random")

    # -- the reference's operators ------------------------------------------------
    def jacobirelaxation(self, level, v, f, mu):
        """PS:125-147: returns v after `mu` smoother sweeps with right-hand side f."""
        self.set_level(level, VEC_U, v)
        self.set_level(level, VEC_B, f)
        self._chk(lib().mgx_smooth(self._h, level, mu), "mgx_smooth")
        return self.get_level(level, VEC_U)

    def residual(self, level, v, f):
        """PS:604-607."""
        self.set_level(level, VEC_U, v)
        self.set_level(level, VEC_B, f)
        self._chk(lib().mgx_residual(self._h, level), "mgx_residual")
        return self.get_level(level, VEC_R)

    def restriction2d(self, level, vec_h):
        """PS:531-546 applied to a fine vector of `level`; returns the level-1 vector."""
        self.set_level(level, VEC_B, vec_h)
        self._chk(lib().mgx_restrict_rhs(self._h, level), "mgx_restrict_rhs")
        return self.get_level(level - 1, VEC_B)

    def residual_restriction(self, level, v, f):
        """PS:604-611 fused: restriction2d(f - A v)."""
        self.set_level(level, VEC_U, v)
        self.set_level(level, VEC_B, f)
        self._chk(lib().mgx_restrict(self._h, level), "mgx_restrict")
        return self.get_level(level - 1, VEC_B), self.get_level(level - 1, VEC_U)

    def interpolation2d(self, level, vec_2h):
        """PS:337-425: coarse vector of level-1 -> fine vector of `level`."""
        self.set_level(level - 1, VEC_U, vec_2h)
        self._chk(lib().mgx_prolong(self._h, level), "mgx_prolong")
        return self.get_level(level, VEC_U)

    def interpolation_add(self, level, vec_h, vec_2h):
        """PS:620-624: vec_h + interpolation2d(vec_2h)."""
        self.set_level(level, VEC_U, vec_h)
        self.set_level(level - 1, VEC_U, vec_2h)
        self._chk(lib().mgx_prolong_add(self._h, level), "mgx_prolong_add")
        return self.get_level(level, VEC_U)

    def bottom_solve(self, f):
        """MF:63-72 / MF:137-139 on the coarsest level."""
        lo = self.cfg.coarsest_level
        self.set_level(lo, VEC_B, f)
        self._chk(lib().mgx_bottom_solve(self._h), "mgx_bottom_solve")
        return self.get_level(lo, VEC_U)

    def vcyclemultigrid(self, level, vec_h, f_h):
        """PS:575-627."""
        self.set_level(level, VEC_U, vec_h)
        self.set_level(level, VEC_B, f_h)
        self._chk(lib().mgx_vcycle(self._h, level), "mgx_vcycle")
        return self.get_level(level, VEC_U)

    def fullmultigrid(self, f_h):
        """PS:629-650 on the finest level."""
        L = self.cfg.finest_level
        self.set_level(L, VEC_B, f_h)
        self._chk(lib().mgx_fmg(self._h), "mgx_fmg")
        return self.get_level(L, VEC_U)

    def residual_norm(self, level=None):
        out = C.c_double()
        lvl = self.cfg.finest_level if level is None else level
        self._chk(lib().mgx_residual_norm(self._h, lvl, C.byref(out)), "mgx_residual_norm")
        return out.value

    def vcycle(self, level=None):
        self._chk(lib().mgx_vcycle(self._h, self.cfg.finest_level if level is None else level), "mgx_vcycle")

    def vcycle_zero(self):
        """one V-cycle from the finest level for A e = b, starting from e = 0"""
        self._chk(lib().mgx_vcycle_zero(self._h), "mgx_vcycle_zero")

    def smooth(self, level, mu):
        self._chk(lib().mgx_smooth(self._h, level, mu), "mgx_smooth")

    def solve(self, tol=1e-8, max_cycles=50):
        """PS:727 run to a tolerance; returns (stats, residual history)."""
        st = Stats()
        hist = np.zeros(max_cycles + 1, dtype=np.float64)
        self._chk(lib().mgx_solve(self._h, tol, max_cycles, C.byref(st), hist.ctypes.data_as(C.POINTER(C.c_double)),
                                  hist.size), "mgx_solve")
        return st, hist[: st.history_len].copy()

    # -- measurement -----------------------------------------------------------------
    def profile_reset(self):
        self._chk(lib().mgx_profile_reset(self._h), "mgx_profile_reset")

    def profile(self):
        p = Profile()
        self._chk(lib().mgx_profile_get(self._h, C.byref(p)), "mgx_profile_get")
        return {"ms": list(p.ms), "launches": list(p.launches), "sweeps": list(p.sweeps)}

    def time_smoother(self, sweeps):
        ms = C.c_double()
        self._chk(lib().mgx_time_smoother(self._h, sweeps, C.byref(ms)), "mgx_time_smoother")
        return ms.value

    def graphs_cached(self):
        """hipGraphs captured by solve() for its loop body (-1: graph replay off)"""
        return lib().mgx_graphs_cached(self._h)

    def synchronize(self):
        self._chk(lib().mgx_synchronize(self._h), "mgx_synchronize")
