// mgx_var.hpp - per-level GENERAL five-point operators: the data model of the reference's second draft
// (Multigrid_functions.cpp = MF) on the structured grid, as gfx950 kernels.
//
//   MF:16-26  ProblemVar { A_sp_dict[level], A_jacobi_sp_dict[level] = {D_inv, R_omega}, b_dict, coarsest_level_matrix }
//   MF:33-41  csr_matrix_elements: CSR per level           -> five coefficient grids per level (c, n, s, w, e)
//   MF:75-96  jacobirelaxation: v <- R_omega v + omega D^-1 b (gemv, gemv alpha = omega, vm::add; MF:86-90)
//                                                          -> k_jacobi_var (one pass instead of three)
//   MF:122-130 restriction2D: injection                    -> k_restrict_inject
//   MF:150-153 residual = f - A v (gemv, vm::sub)          -> k_residual_var
//   MF:63-72, 137-139 direct solve of the coarsest system  -> dense inverse by Gauss-Jordan (set-up), matvec per solve
// (A u)_ij = c u_ij + n u_(i-1)j + s u_(i+1)j + w u_i(j-1) + e u_i(j+1); the Dirichlet ring of the grid
// layout is zero, so boundary-adjacent points need no special case.  A CSR row of a row-major five-point
// operator lists its columns as N, W, C, E, S, and that is the order every sum below is taken in; no
// contraction (-ffp-contract=off): bit-identical to the oracle's restatement (oracle/mg_oracle_var.inc).
//
// Roofline: every coefficient is read once per sweep, so a sweep moves v, b, D_inv, R_n, R_s, R_w, R_e in
// and v' out: 8 sizeof(T) per point (64 B in double against the constant stencil's 24 B); the residual
// moves v, b, c, n, s, w, e in and r out, also 8 sizeof(T).  Both are single passes of independent rows
// (one wave per row and strip, like k_jacobi_rows): HBM-bound, no temporal fusion (a K-level pass would
// need K-row windows of five more arrays in registers).
#pragma once

#include "mgx_kernels.hpp"

namespace mgx {

template <typename T> struct Lanes { T a[VecOf<T>::W]; };
__device__ __forceinline__ Lanes<double> to_lanes(const double2& v) { return Lanes<double>{{v.x, v.y}}; }
__device__ __forceinline__ Lanes<float> to_lanes(const float4& v) { return Lanes<float>{{v.x, v.y, v.z, v.w}}; }
__device__ __forceinline__ double2 from_lanes(const Lanes<double>& l) { return make_double2(l.a[0], l.a[1]); }
__device__ __forceinline__ float4 from_lanes(const Lanes<float>& l) { return make_float4(l.a[0], l.a[1], l.a[2], l.a[3]); }

// sum_k coef_k * neighbour_k in CSR column order N, W, C, E, S; `centre` is the coefficient of the point itself
// (an array element for A, the scalar 1 - omega for R_omega)
template <typename T, typename CF>
__device__ __forceinline__ Lanes<T> stencil5(const Lanes<T>& up, const Lanes<T>& cur, const Lanes<T>& dn, T left, T right,
                                             const Lanes<T>& cn, const Lanes<T>& cw, CF centre, const Lanes<T>& ce, const Lanes<T>& cs)
{
    constexpr int W = VecOf<T>::W;
    Lanes<T> o;
#pragma unroll
    for (int k = 0; k < W; ++k) {
        const T l = (k == 0) ? left : cur.a[k - 1];
        const T r = (k == W - 1) ? right : cur.a[k + 1];
        T acc = cn.a[k] * up.a[k];
        acc = acc + cw.a[k] * l;
        acc = acc + centre(k) * cur.a[k];
        acc = acc + ce.a[k] * r;
        acc = acc + cs.a[k] * dn.a[k];
        o.a[k] = acc;
    }
    return o;
}

// MF:75-96: one sweep of v' = R_omega v + omega (D_inv b), out of place; rows [row_lo, row_hi)
template <typename T>
__global__ void __launch_bounds__(kBlock)
k_jacobi_var(const T* __restrict__ vin, const T* __restrict__ rhs, T* __restrict__ vout, const T* __restrict__ dinv,
             const T* __restrict__ rn, const T* __restrict__ rs, const T* __restrict__ rw, const T* __restrict__ re,
             int N, long pitch, int row_lo, int row_hi, int strips, T rc, T omega, int rows_alloc)
{
    using V = typename VecOf<T>::type;
    constexpr int W = VecOf<T>::W;
    const Tile t = wave_tile(strips, row_hi - row_lo);
    if (!t.active) return;
    const Cols c = lane_cols<W>(t.strip, N, pitch);
    const int r = row_lo + t.chunk;
    const long at = c.col + (long)r * pitch;
    const bool in = c.ld && r >= 0 && r < rows_alloc;
    const V up = vload<V>(vin + at - pitch, c.ld && r >= 1 && r <= rows_alloc);
    const V cur = vload<V>(vin + at, in);
    const V dn = vload<V>(vin + at + pitch, c.ld && r >= -1 && r + 1 < rows_alloc);
    const V bb = vload<V>(rhs + at, in);
    const V dv = vload<V>(dinv + at, in);
    const Lanes<T> n = to_lanes(vload<V>(rn + at, in)), s = to_lanes(vload<V>(rs + at, in));
    const Lanes<T> w = to_lanes(vload<V>(rw + at, in)), e = to_lanes(vload<V>(re + at, in));
    const T left = from_left(last(cur)), right = from_right(first(cur));
    const Lanes<T> p1 = stencil5<T>(to_lanes(up), to_lanes(cur), to_lanes(dn), left, right, n, w, [&](int) { return rc; }, e, s);   // MF:86
    const Lanes<T> b = to_lanes(bb), d = to_lanes(dv);
    Lanes<T> o;
#pragma unroll
    for (int k = 0; k < W; ++k) o.a[k] = p1.a[k] + omega * (d.a[k] * b.a[k]);                                                   // MF:88, 90
    V ov = from_lanes(o);
    mask_cols(ov, c.col, N);
    vstore<V>(vout + at, ov, c.st && in);
}

// MF:150-153: r = b - A v.  MODE 0: store r;  MODE 1: per-block sums of r^2 (the norm the solve reports)
template <typename T, int MODE>
__global__ void __launch_bounds__(kBlock)
k_residual_var(const T* __restrict__ vin, const T* __restrict__ rhs, T* __restrict__ out, double* __restrict__ partial,
               const T* __restrict__ ac, const T* __restrict__ an, const T* __restrict__ as, const T* __restrict__ aw,
               const T* __restrict__ ae, int N, long pitch, int row_lo, int row_hi, int strips, int rows_alloc)
{
    using V = typename VecOf<T>::type;
    constexpr int W = VecOf<T>::W;
    __shared__ double wsum[kWavesPerBlock];
    const Tile t = wave_tile(strips, row_hi - row_lo);
    double acc = 0.0;
    if (t.active) {
        const Cols c = lane_cols<W>(t.strip, N, pitch);
        const int r = row_lo + t.chunk;
        const long at = c.col + (long)r * pitch;
        const bool in = c.ld && r >= 0 && r < rows_alloc;
        const V up = vload<V>(vin + at - pitch, c.ld && r >= 1 && r <= rows_alloc);
        const V cur = vload<V>(vin + at, in);
        const V dn = vload<V>(vin + at + pitch, c.ld && r >= -1 && r + 1 < rows_alloc);
        const Lanes<T> b = to_lanes(vload<V>(rhs + at, in));
        const Lanes<T> cc = to_lanes(vload<V>(ac + at, in));
        const Lanes<T> n = to_lanes(vload<V>(an + at, in)), s = to_lanes(vload<V>(as + at, in));
        const Lanes<T> w = to_lanes(vload<V>(aw + at, in)), e = to_lanes(vload<V>(ae + at, in));
        const T left = from_left(last(cur)), right = from_right(first(cur));
        const Lanes<T> av = stencil5<T>(to_lanes(up), to_lanes(cur), to_lanes(dn), left, right, n, w, [&](int k) { return cc.a[k]; }, e, s);
        Lanes<T> o;
#pragma unroll
        for (int k = 0; k < W; ++k) o.a[k] = b.a[k] - av.a[k];
        V ov = from_lanes(o);
        mask_cols(ov, c.col, N);
        if (MODE == 0) {
            vstore<V>(out + at, ov, c.st && in);
        } else if (c.st && in) {
            const Lanes<T> q = to_lanes(ov);
            if constexpr (W == 2) acc = (double)q.a[0] * (double)q.a[0] + (double)q.a[1] * (double)q.a[1];
            else acc = ((double)q.a[0] * (double)q.a[0] + (double)q.a[1] * (double)q.a[1]) +
                       ((double)q.a[2] * (double)q.a[2] + (double)q.a[3] * (double)q.a[3]);
        }
    }
    if (MODE != 0) {
        for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, kWave);
        if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = acc;
        __syncthreads();
        if (threadIdx.x == 0) {
            double sum = 0.0;
            for (int w2 = 0; w2 < kWavesPerBlock; ++w2) sum += wsum[w2];
            partial[blockIdx.x] = sum;
        }
    }
}

// MF:122-130: coarse(I, J) = wgt * fine(2I, 2J); optionally zero the coarse guess (PS:613) in the same pass
template <typename T>
__global__ void __launch_bounds__(kBlock)
k_restrict_inject(const T* __restrict__ fine, T* __restrict__ coarse, T* __restrict__ coarse_zero, int NC, long pitch,
                  long cpitch, T wgt)
{
    const int J = blockIdx.x * blockDim.x + threadIdx.x;
    const int I = blockIdx.y + 1;
    if (J < 1 || J >= NC || I >= NC) return;
    coarse[(long)I * cpitch + J] = wgt * fine[(long)(2 * I) * pitch + 2 * J];
    if (coarse_zero) coarse_zero[(long)I * cpitch + J] = (T)0;
}

// A_jacobi_sp_dict[level] from A_sp_dict[level] (MF:28-32): D_inv = 1 / c, R_x = -(omega (D_inv a_x)); the
// diagonal of R_omega is 1 - omega exactly (D^-1 A has a unit diagonal) and is not stored
template <typename T>
__global__ void k_var_build_jacobi(const T* __restrict__ ac, const T* __restrict__ an, const T* __restrict__ as,
                                   const T* __restrict__ aw, const T* __restrict__ ae, T* __restrict__ dinv,
                                   T* __restrict__ rn, T* __restrict__ rs, T* __restrict__ rw, T* __restrict__ re,
                                   int N, long pitch, T omega)
{
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    const int r = blockIdx.y;
    if (c < 1 || c >= N || r < 1 || r >= N) return;
    const long at = (long)r * pitch + c;
    const T d = (T)1 / ac[at];
    dinv[at] = d;
    rn[at] = -(omega * (d * an[at]));
    rs[at] = -(omega * (d * as[at]));
    rw[at] = -(omega * (d * aw[at]));
    re[at] = -(omega * (d * ae[at]));
}

// -div(a grad u) on a level from the nodal coefficient of the finest grid (rows 0..Nf of Nf + 1 doubles),
// sampled at the level's nodes (stride q); face coefficient = mean of its two nodes (oracle: stencil_from_nodes)
template <typename T>
__global__ void k_var_from_nodes(const double* __restrict__ a, int Nf, int q, T* __restrict__ ac, T* __restrict__ an,
                                 T* __restrict__ as, T* __restrict__ aw, T* __restrict__ ae, int N, long pitch)
{
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    const int r = blockIdx.y;
    if (c < 1 || c >= N || r < 1 || r >= N) return;
    const long ld = (long)Nf + 1;
    auto node = [&](int i, int j) { return a[(long)(i * q) * ld + (long)(j * q)]; };
    const double ctr = node(r, c);
    const double fn = 0.5 * (ctr + node(r - 1, c)), fs = 0.5 * (ctr + node(r + 1, c));
    const double fw = 0.5 * (ctr + node(r, c - 1)), fe = 0.5 * (ctr + node(r, c + 1));
    const long at = (long)r * pitch + c;
    ac[at] = (T)(((fn + fw) + fe) + fs);
    an[at] = (T)(-fn); as[at] = (T)(-fs); aw[at] = (T)(-fw); ae[at] = (T)(-fe);
}

// ---- direct bottom solve of a general coarsest operator (MF:63-72, 137-139) ---------------------------------
// Dense inverse of the n^2 x n^2 matrix by Gauss-Jordan elimination without pivoting (diagonally dominant
// M-matrices), in double whatever the hierarchy's type: two launches per pivot at set-up time, one matvec per
// solve.  Every element update is its own IEEE operations in the oracle's order (orc_dense_inverse): same bits.
template <typename T>
__global__ void k_var_dense_fill(double* __restrict__ M, double* __restrict__ Inv, const T* __restrict__ ac,
                                 const T* __restrict__ an, const T* __restrict__ as, const T* __restrict__ aw,
                                 const T* __restrict__ ae, int n, long pitch)
{
    const int NN = n * n;
    const int j = blockIdx.x * blockDim.x + threadIdx.x;      // column
    const int k = blockIdx.y;                                  // row = unknown (ri, rj)
    if (j >= NN) return;
    const int ri = k / n, rj = k - ri * n;
    const long at = (long)(ri + 1) * pitch + (rj + 1);
    double v = 0.0;
    if (j == k) v = (double)ac[at];
    else if (j == k - n) v = (double)an[at];
    else if (j == k + n) v = (double)as[at];
    else if (j == k - 1 && rj > 0) v = (double)aw[at];
    else if (j == k + 1 && rj < n - 1) v = (double)ae[at];
    M[(long)k * NN + j] = v;
    Inv[(long)k * NN + j] = (j == k) ? 1.0 : 0.0;
}
static __global__ void k_gj_prow(const double* __restrict__ M, const double* __restrict__ Inv, double* __restrict__ pm,
                                 double* __restrict__ pi, int NN, int k)
{
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= NN) return;
    const double p = M[(long)k * NN + k];
    pm[j] = M[(long)k * NN + j] / p;
    pi[j] = Inv[(long)k * NN + j] / p;
}
// one workgroup per row: every thread reads the row's multiplier before any thread overwrites it
static __global__ void k_gj_elim(double* __restrict__ M, double* __restrict__ Inv, const double* __restrict__ pm,
                                 const double* __restrict__ pi, int NN, int k)
{
    const int i = blockIdx.x;
    double* mi = M + (long)i * NN;
    double* ii = Inv + (long)i * NN;
    const double f = mi[k];
    __syncthreads();
    for (int j = threadIdx.x; j < NN; j += blockDim.x) {
        if (i == k) { mi[j] = pm[j]; ii[j] = pi[j]; }
        else { mi[j] = mi[j] - f * pm[j]; ii[j] = ii[j] - f * pi[j]; }
    }
}
// x = Inv b on the coarsest grid: one thread per unknown, in-order row sum (deterministic, n^2 <= 961 terms)
template <typename T>
__global__ void k_var_dense_solve(const double* __restrict__ Inv, const T* __restrict__ b, T* __restrict__ x, int n, long pitch)
{
    const int NN = n * n;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= NN) return;
    const double* row = Inv + (long)i * NN;
    double acc = 0.0;
    for (int q = 0; q < NN; ++q) {
        const int bi = q / n, bj = q - bi * n;
        acc += row[q] * (double)b[(long)(bi + 1) * pitch + (bj + 1)];
    }
    const int ri = i / n, rj = i - ri * n;
    x[(long)(ri + 1) * pitch + (rj + 1)] = (T)acc;
}

} // namespace mgx
