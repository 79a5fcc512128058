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
// n <= 255, so the four n^3 products are ~8 MFLOP: two small launches of two products each, always
// in double whatever the level's storage type.  This is a direct solver (no
// iteration, no tolerance); the oracle uses banded Cholesky, an independent
// exact method, so agreement between the two checks both.
#pragma once

#include <hip/hip_runtime.h>
#include <cmath>
#include <vector>

#ifndef MGX_DST_DEEP
#define MGX_DST_DEEP 32
#endif

namespace mgx {

// Two of the four products per launch.  U = S ((S B S) ./ lambda) S is two row-local pairs - row i of (S B) S needs
// only row i of S B, row i of (S T) S only row i of S T - so one workgroup owns output row i: its threads j first form
// P[i][j] = sum_k S[i][k] X[k][j] (X the level's grid b, or the scaled transform T of the first launch), park the row
// in LDS, and then form sum_k P[i][k] S[k][j].  Every sum runs in k order with one accumulator, exactly the four
// products of the oracle's ORC_BOTTOM_DST (same bits as the four-launch form of rounds 1-2: 4 x 9 us -> 2 launches).
//  FIRST: X is the padded grid of type T (element (k, j) at (k + 1, j + 1)); the result is divided by
//         s[i] + s[j] and written to the n x n double array `out`
//  else : X is the n x n double array; the result is written to the padded grid of type T
template <typename T, bool FIRST>
__global__ void __launch_bounds__(256)
k_dst_pair(const double* __restrict__ S, const void* __restrict__ Xv, void* __restrict__ Ov, const double* __restrict__ s,
           int n, long gpitch)
{
    __shared__ double prow[256];
    const int j = threadIdx.x;
    const int i = blockIdx.x;                       // wave-uniform: row i of S goes through the scalar cache
    const bool live = j < n;
    // the loads of a sum are independent of its accumulator: they are issued kDeep at a time (a 127-term dot
    // product is nothing but L2 latency otherwise; 64 deep measured slower: 98 against 86 us for the coarse part of the reference hierarchy)
    constexpr int kDeep = MGX_DST_DEEP;
    const double* Sr = S + (long)i * n;
    double acc = 0.0;
    {
        int k = 0;
        for (; k + kDeep <= n; k += kDeep) {
            double x[kDeep];
#pragma unroll
            for (int q = 0; q < kDeep; ++q) {
                if constexpr (FIRST) x[q] = live ? (double)(reinterpret_cast<const T*>(Xv)[(long)(k + q + 1) * gpitch + (j + 1)]) : 0.0;
                else x[q] = live ? reinterpret_cast<const double*>(Xv)[(long)(k + q) * n + j] : 0.0;
            }
#pragma unroll
            for (int q = 0; q < kDeep; ++q) acc += Sr[k + q] * x[q];
        }
        double x[kDeep];
#pragma unroll
        for (int q = 0; q < kDeep; ++q) {
            const bool ok = live && (k + q < n);
            if constexpr (FIRST) x[q] = ok ? (double)(reinterpret_cast<const T*>(Xv)[(long)(k + q + 1) * gpitch + (j + 1)]) : 0.0;
            else x[q] = ok ? reinterpret_cast<const double*>(Xv)[(long)(k + q) * n + j] : 0.0;
        }
#pragma unroll
        for (int q = 0; q < kDeep; ++q)
            if (k + q < n) acc += Sr[k + q] * x[q];
    }
    prow[j] = acc;
    __syncthreads();
    double out = 0.0;
    {
        int k = 0;
        for (; k + kDeep <= n; k += kDeep) {
            double x[kDeep];
#pragma unroll
            for (int q = 0; q < kDeep; ++q) x[q] = live ? S[(long)(k + q) * n + j] : 0.0;
#pragma unroll
            for (int q = 0; q < kDeep; ++q) out += prow[k + q] * x[q];
        }
        double x[kDeep];
#pragma unroll
        for (int q = 0; q < kDeep; ++q) x[q] = (live && k + q < n) ? S[(long)(k + q) * n + j] : 0.0;
#pragma unroll
        for (int q = 0; q < kDeep; ++q)
            if (k + q < n) out += prow[k + q] * x[q];
    }
    if (!live) return;
    if constexpr (FIRST) reinterpret_cast<double*>(Ov)[(long)i * n + j] = out / (s[i] + s[j]);
    else reinterpret_cast<T*>(Ov)[(long)(i + 1) * gpitch + (j + 1)] = (T)out;
}

struct BottomDST {
    int n = 0;
    double* S = nullptr;      // n x n sine matrix
    double* s = nullptr;      // n eigenvalue halves: 4 sin^2(pi (i+1) / (2 (n+1)))
    double* w1 = nullptr;     // n x n: (S B S) ./ lambda between the two launches

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
        if ((e = hipMemcpy(S, hS.data(), nn * sizeof(double), hipMemcpyHostToDevice)) != hipSuccess) return e;
        if ((e = hipMemcpy(s, hs.data(), n * sizeof(double), hipMemcpyHostToDevice)) != hipSuccess) return e;
        return hipSuccess;
    }

    void destroy()
    {
        if (S) (void)hipFree(S);
        if (s) (void)hipFree(s);
        if (w1) (void)hipFree(w1);
        S = s = w1 = nullptr;
    }

    // u_grid = A^-1 b_grid on the padded coarsest-level grids (2 launches; w1 holds (S B S) ./ lambda in between)
    template <typename T>
    void solve(const T* b_grid, T* u_grid, long gpitch, hipStream_t st) const
    {
        hipLaunchKernelGGL((k_dst_pair<T, true>), dim3(n), dim3(256), 0, st, S, (const void*)b_grid, (void*)w1, s, n, gpitch);
        hipLaunchKernelGGL((k_dst_pair<T, false>), dim3(n), dim3(256), 0, st, S, (const void*)w1, (void*)u_grid, s, n, gpitch);
    }
};

} // namespace mgx
