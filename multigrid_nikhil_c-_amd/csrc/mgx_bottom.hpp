// mgx_bottom.hpp — exact coarsest-level solve on the GPU.
//
// Replaces MF:63-72 direct_solver (Eigen SparseLU on the host) called from
// MF:137-139 and MF:178-181; the first draft's PS:581-587 "solve" (mu1+mu2
// Jacobi sweeps) is kept only as MGX_BOTTOM_SMOOTH.
//
// The 5-point Dirichlet Laplacian on an n x n grid is diagonalised by the
// type-I discrete sine transform: with S_jk = sqrt(2/(n+1)) sin(pi j k/(n+1))
// (symmetric, S S = I) and lambda_ij = 4 sin^2(pi i/2(n+1)) + 4 sin^2(pi j/2(n+1)),
//     U = S ( (S B S) ./ lambda ) S .
// n <= 255, so the four n^3 products are ~8 MFLOP: four small launches, always
// in double whatever the level's storage type.  This is a direct solver (no
// iteration, no tolerance); the oracle uses banded Cholesky, an independent
// exact method, so agreement between the two checks both.
#pragma once

#include <hip/hip_runtime.h>
#include <cmath>
#include <vector>

namespace mgx {

// C = op(A * B): one thread per output, k-loop in registers.
//  IN_GRID : B (step 1) is the level's padded grid of type T, element (k,j) at (k+1, j+1)
//  OUT_GRID: C (step 4) is written to the padded grid of type T
//  SCALE   : multiply by 1 / (s[i] + s[j])   (step 2)
template <typename T, bool IN_GRID, bool OUT_GRID, bool SCALE>
__global__ void __launch_bounds__(256)
k_dst_gemm(const double* __restrict__ A, const void* __restrict__ Bv, void* __restrict__ Cv,
           const double* __restrict__ s, int n, long gpitch)
{
    const int j = blockIdx.x * 64 + (threadIdx.x & 63);
    // the row index is the same for a whole wave: as a scalar, row i of A is read through the scalar
    // cache (s_load) and costs no vector memory instruction
    const int i = blockIdx.y * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (i >= n || j >= n) return;
    // The sum runs in k order with one accumulator (deterministic); the loads are independent of
    // it, so they are issued kDeep at a time: a 127-term dot product is nothing but L2 latency
    // (8 deep: 16 round trips, 10.5 us per product; 32 deep: 4).
    constexpr int kDeep = 32;
    double acc = 0.0;
    const double* Ar = A + (long)i * n;
    int k = 0;
    if (IN_GRID) {
        const T* B = reinterpret_cast<const T*>(Bv) + gpitch + (j + 1);
        for (; k + kDeep <= n; k += kDeep) {
            double b[kDeep];
#pragma unroll
            for (int q = 0; q < kDeep; ++q) b[q] = (double)B[(long)(k + q) * gpitch];
#pragma unroll
            for (int q = 0; q < kDeep; ++q) acc += Ar[k + q] * b[q];
        }
        {
            // the remainder, fetched together as well (rows beyond n - 1 are not read)
            double b[kDeep];
#pragma unroll
            for (int q = 0; q < kDeep; ++q) b[q] = (k + q < n) ? (double)B[(long)(k + q) * gpitch] : 0.0;
#pragma unroll
            for (int q = 0; q < kDeep; ++q)
                if (k + q < n) acc += Ar[k + q] * b[q];
        }
    } else {
        const double* B = reinterpret_cast<const double*>(Bv) + j;
        for (; k + kDeep <= n; k += kDeep) {
            double b[kDeep];
#pragma unroll
            for (int q = 0; q < kDeep; ++q) b[q] = B[(long)(k + q) * n];
#pragma unroll
            for (int q = 0; q < kDeep; ++q) acc += Ar[k + q] * b[q];
        }
        {
            double b[kDeep];
#pragma unroll
            for (int q = 0; q < kDeep; ++q) b[q] = (k + q < n) ? B[(long)(k + q) * n] : 0.0;
#pragma unroll
            for (int q = 0; q < kDeep; ++q)
                if (k + q < n) acc += Ar[k + q] * b[q];
        }
    }
    if (SCALE) acc = acc / (s[i] + s[j]);
    if (OUT_GRID) reinterpret_cast<T*>(Cv)[(long)(i + 1) * gpitch + (j + 1)] = (T)acc;
    else reinterpret_cast<double*>(Cv)[(long)i * n + j] = acc;
}

struct BottomDST {
    int n = 0;
    double* S = nullptr;      // n x n sine matrix
    double* s = nullptr;      // n eigenvalue halves: 4 sin^2(pi (i+1) / (2 (n+1)))
    double* w1 = nullptr;     // n x n work
    double* w2 = nullptr;

    hipError_t init(int n_)
    {
        n = n_;
        const size_t nn = (size_t)n * n;
        std::vector<double> hS(nn), hs(n);
        const long double pi = 3.14159265358979323846264338327950288L;
        const long double norm = sqrtl(2.0L / (long double)(n + 1));
        for (int i = 0; i < n; ++i) {
            const long double a = sinl(pi * (long double)(i + 1) / (2.0L * (long double)(n + 1)));
            hs[i] = (double)(4.0L * a * a);
            for (int k = 0; k < n; ++k) {
                // reduce the argument exactly before calling sin
                const long m = ((long)(i + 1) * (long)(k + 1)) % (2L * (n + 1));
                hS[(size_t)i * n + k] = (double)(norm * sinl(pi * (long double)m / (long double)(n + 1)));
            }
        }
        hipError_t e;
        if ((e = hipMalloc(&S, nn * sizeof(double))) != hipSuccess) return e;
        if ((e = hipMalloc(&s, n * sizeof(double))) != hipSuccess) return e;
        if ((e = hipMalloc(&w1, nn * sizeof(double))) != hipSuccess) return e;
        if ((e = hipMalloc(&w2, nn * sizeof(double))) != hipSuccess) return e;
        if ((e = hipMemcpy(S, hS.data(), nn * sizeof(double), hipMemcpyHostToDevice)) != hipSuccess) return e;
        if ((e = hipMemcpy(s, hs.data(), n * sizeof(double), hipMemcpyHostToDevice)) != hipSuccess) return e;
        return hipSuccess;
    }

    void destroy()
    {
        if (S) (void)hipFree(S);
        if (s) (void)hipFree(s);
        if (w1) (void)hipFree(w1);
        if (w2) (void)hipFree(w2);
        S = s = w1 = w2 = nullptr;
    }

    // u_grid = A^-1 b_grid on the padded coarsest-level grids (4 launches)
    template <typename T>
    void solve(const T* b_grid, T* u_grid, long gpitch, hipStream_t st) const
    {
        const dim3 blk(256);
        const dim3 grd((n + 63) / 64, (n + 3) / 4);
        hipLaunchKernelGGL((k_dst_gemm<T, true, false, false>), grd, blk, 0, st, S, (const void*)b_grid, (void*)w1, s, n, gpitch);
        hipLaunchKernelGGL((k_dst_gemm<T, false, false, true>), grd, blk, 0, st, w1, (const void*)S, (void*)w2, s, n, gpitch);
        hipLaunchKernelGGL((k_dst_gemm<T, false, false, false>), grd, blk, 0, st, S, (const void*)w2, (void*)w1, s, n, gpitch);
        hipLaunchKernelGGL((k_dst_gemm<T, false, true, false>), grd, blk, 0, st, w1, (const void*)S, (void*)u_grid, s, n, gpitch);
    }
};

} // namespace mgx
