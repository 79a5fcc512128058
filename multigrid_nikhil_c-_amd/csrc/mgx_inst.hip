// mgx_inst.hip - explicit instantiations of the smoother launch wrappers (and through them of the
// k_jacobi_cycle / k_jacobi_fused / k_tile_smooth kernels), one group per compilation so that the library
// builds in parallel (see the end of mgx_launch.hpp and the Makefile):
//   -DMGX_INST_KIND=1  launch_cycle<T, PRE, POST, SM, AR> for the (PRE, POST) pair number MGX_INST_PP
//   -DMGX_INST_KIND=2  launch_fused<T, SM, AR>
//   -DMGX_INST_KIND=3  smooth_tiled<T, SM, AR>  (all six (PRE, POST) pairs of k_tile_smooth)
//   -DMGX_INST_T=double|float   -DMGX_INST_SM=0|1   -DMGX_INST_AR=0|1
#include "mgx_launch.hpp"

namespace mgx {
using T_ = MGX_INST_T;
#if MGX_INST_KIND == 1
#if MGX_INST_PP == 0
#define PP_ 1, 2
#elif MGX_INST_PP == 1
#define PP_ 1, 0
#elif MGX_INST_PP == 2
#define PP_ 0, 1
#elif MGX_INST_PP == 3
#define PP_ 0, 2
#else
#define PP_ 1, 1
#endif
template int launch_cycle<T_, PP_, MGX_INST_SM, MGX_INST_AR>(int, const T_*, const T_*, T_*, const FoldArgs&, int, long, T_, T_, int, hipStream_t);
#elif MGX_INST_KIND == 2
template bool launch_fused<T_, MGX_INST_SM, MGX_INST_AR>(int, const T_*, const T_*, T_*, int, long, int, int, T_, T_, int, int, int, int, hipStream_t, int, int);
#elif MGX_INST_KIND == 3
template int smooth_tiled<T_, MGX_INST_SM, MGX_INST_AR>(T_*, const T_*, T_*, int, long, int, double, int, FoldArgs, bool, int, bool, hipStream_t, int*);
#else
#error "MGX_INST_KIND must be 1, 2 or 3"
#endif
} // namespace mgx
