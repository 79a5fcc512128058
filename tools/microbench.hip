// tools/microbench.hip — kernel-variant microbenchmarks on one MI355X.
// Not part of the product: a measuring stick for DESIGN.md's kernel choices.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/microbench.hip -o tools/microbench
//   ./tools/microbench [level=13] [iters=20]
// Prints, per kernel variant, average launch time and algorithmic GB/s
// (SURVEY §8d byte counts) next to a same-traffic streaming ceiling
// (2 reads + 1 write of the same arrays with no stencil).
#include "../multigrid_nikhil_c-_amd/csrc/mgx_kernels.hpp"

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <functional>
#include <type_traits>
#include <vector>

using namespace mgx;

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1); } } while (0)

// ---- streaming ceiling with the Jacobi traffic shape: out = a + b over the padded grid ----
template <typename T>
__global__ void __launch_bounds__(256) k_stream3(const T* __restrict__ a, const T* __restrict__ b, T* __restrict__ o, long nvec)
{
    using V = typename VecOf<T>::type;
    const V* pa = reinterpret_cast<const V*>(a);
    const V* pb = reinterpret_cast<const V*>(b);
    V* po = reinterpret_cast<V*>(o);
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += (long)gridDim.x * blockDim.x) {
        V x = pa[i], y = pb[i];
        x.x += y.x; x.y += y.y;
        if constexpr (VecOf<T>::W == 4) { x.z += y.z; x.w += y.w; }
        po[i] = x;
    }
}

// ---- variant N: no marching; one wave handles 62 vectors of ONE row (relies on L2 for row reuse) ----
template <typename T>
__global__ void __launch_bounds__(kBlock)
k_jacobi_naive(const T* __restrict__ vin, const T* __restrict__ rhs, T* __restrict__ vout,
               int N, long pitch, int row_lo, int row_hi, int strips, T c0, T c1)
{
    using V = typename VecOf<T>::type;
    constexpr int W = VecOf<T>::W;
    const Tile t = wave_tile(strips, row_hi - row_lo);
    if (!t.active) return;
    const Cols c = lane_cols<W>(t.strip, N, pitch);
    const int r = row_lo + t.chunk;
    const T* pv = vin + c.col + (long)r * pitch;
    const V up = vload<V>(pv - pitch, c.ld);
    const V cur = vload<V>(pv, c.ld);
    const V dn = vload<V>(pv + pitch, c.ld);
    const V bb = vload<V>(rhs + c.col + (long)r * pitch, c.ld);
    V o = jacobi_vec<T>(up, cur, dn, bb, c0, c1);
    if (c.vx == 0) o.x = (T)0;
    vstore<V>(vout + c.col + (long)r * pitch, o, c.st);
}

// ---- variant B: 64 storing lanes per wave, line-aligned strips; the two edge lanes fetch
//      the one halo element they need with a single extra (2-lane) load per row ----
template <typename T>
__global__ void __launch_bounds__(kBlock)
k_jacobi_b(const T* __restrict__ vin, const T* __restrict__ rhs, T* __restrict__ vout,
           int N, long pitch, int row_lo, int row_hi, int R, int strips, int chunks, T c0, T c1)
{
    using V = typename VecOf<T>::type;
    constexpr int W = VecOf<T>::W;
    const Tile t = wave_tile(strips, chunks);
    if (!t.active) return;
    const int lane = threadIdx.x & 63;
    const int vx = t.strip * 64 + lane;
    const long col = (long)vx * W;
    const bool ld = (col + W <= pitch);
    const bool st = vx < N / W;
    const int r0 = row_lo + t.chunk * R;
    const int r1 = min(r0 + R, row_hi);
    const T* pv = vin + col;
    const T* pb = rhs + col;
    T* po = vout + col;
    // halo element address offset relative to pv (row start): lane 0 -> col-1, lane 63 -> col+W
    const long hoff = (lane == 0) ? -1 : W;
    const bool hld = (lane == 0 && col > 0) || (lane == 63 && col + W < pitch);
    auto halo = [&](int row) -> T { T h = (T)0; if (hld) h = pv[(long)row * pitch + hoff]; return h; };

    V up = vload<V>(pv + (long)(r0 - 1) * pitch, ld);
    V cur = vload<V>(pv + (long)r0 * pitch, ld);
    T hcur = halo(r0);
    for (int r = r0; r < r1; ++r) {
        const V dn = vload<V>(pv + (long)(r + 1) * pitch, ld);
        const T hdn = halo(r + 1);
        const V bb = vload<V>(pb + (long)r * pitch, ld);
        T l = from_left(last(cur)), rr = from_right(first(cur));
        if (lane == 0) l = hcur;
        if (lane == 63) rr = hcur;
        V o;
        if constexpr (W == 2) {
            o.x = (c0 * cur.x + c1 * bb.x) + c1 * nbr(up.x, l, cur.y, dn.x);
            o.y = (c0 * cur.y + c1 * bb.y) + c1 * nbr(up.y, cur.x, rr, dn.y);
        } else {
            o.x = (c0 * cur.x + c1 * bb.x) + c1 * nbr(up.x, l, cur.y, dn.x);
            o.y = (c0 * cur.y + c1 * bb.y) + c1 * nbr(up.y, cur.x, cur.z, dn.y);
            o.z = (c0 * cur.z + c1 * bb.z) + c1 * nbr(up.z, cur.y, cur.w, dn.z);
            o.w = (c0 * cur.w + c1 * bb.w) + c1 * nbr(up.w, cur.z, rr, dn.w);
        }
        if (vx == 0) o.x = (T)0;
        vstore<V>(po + (long)r * pitch, o, st);
        up = cur; cur = dn; hcur = hdn;
    }
}

static bool g_fused_only = false;

struct Timer {
    hipEvent_t a, b;
    Timer() { CK(hipEventCreate(&a)); CK(hipEventCreate(&b)); }
    float run(const std::function<void()>& f, int warm, int iters)
    {
        for (int i = 0; i < warm; ++i) f();
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(a));
        for (int i = 0; i < iters; ++i) f();
        CK(hipEventRecord(b));
        CK(hipEventSynchronize(b));
        float ms = 0;
        CK(hipEventElapsedTime(&ms, a, b));
        CK(hipGetLastError());
        return ms / iters;
    }
};

template <typename T>
__global__ void k_init(T* p, long n, uint32_t seed)
{
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        uint32_t x = (uint32_t)i * 2654435761u + seed;
        x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15;
        p[i] = (T)((double)(x & 0xFFFFFF) / 8388608.0 - 1.0);
    }
}

template <typename T>
void run_level(int level, int iters)
{
    constexpr int W = VecOf<T>::W;
    const int N = 1 << level;
    const long align = 256 / sizeof(T);
    const long pitch = (N + 1 + align - 1) / align * align;
    const long elems = (long)(N + 1) * pitch;
    const size_t bytes = elems * sizeof(T);
    const int NC = N / 2;
    const long cpitch = (NC + 1 + align - 1) / align * align;
    const size_t cbytes = (size_t)(NC + 1) * cpitch * sizeof(T);
    T *u, *b, *tmp, *cb, *cu;
    CK(hipMalloc(&u, bytes)); CK(hipMalloc(&b, bytes)); CK(hipMalloc(&tmp, bytes));
    CK(hipMalloc(&cb, cbytes)); CK(hipMalloc(&cu, cbytes));
    // random data everywhere except the ring/padding (zero)
    CK(hipMemset(u, 0, bytes)); CK(hipMemset(b, 0, bytes)); CK(hipMemset(tmp, 0, bytes));
    CK(hipMemset(cb, 0, cbytes)); CK(hipMemset(cu, 0, cbytes));
    {
        T* host = nullptr; (void)host;
        // fill interiors through a strided init: simplest is init all, then zero ring via memset2D
        hipLaunchKernelGGL(k_init<T>, dim3(2048), dim3(256), 0, 0, u, elems, 1u);
        hipLaunchKernelGGL(k_init<T>, dim3(2048), dim3(256), 0, 0, b, elems, 2u);
        hipLaunchKernelGGL(k_init<T>, dim3(1024), dim3(256), 0, 0, cu, (long)(NC + 1) * cpitch, 3u);
        CK(hipMemset(u, 0, pitch * sizeof(T)));                                   // row 0
        CK(hipMemset(u + (long)N * pitch, 0, pitch * sizeof(T)));                 // row N
        CK(hipMemset2D(u, pitch * sizeof(T), 0, sizeof(T), N + 1));               // col 0
        CK(hipMemset2D(u + N, pitch * sizeof(T), 0, (pitch - N) * sizeof(T), N + 1));   // cols >= N
        CK(hipMemset(cu, 0, cpitch * sizeof(T)));
        CK(hipMemset(cu + (long)NC * cpitch, 0, cpitch * sizeof(T)));
        CK(hipMemset2D(cu, cpitch * sizeof(T), 0, sizeof(T), NC + 1));
        CK(hipMemset2D(cu + NC, cpitch * sizeof(T), 0, (cpitch - NC) * sizeof(T), NC + 1));
    }
    CK(hipDeviceSynchronize());
    Timer tm;
    const double pts = (double)(N - 1) * (double)(N - 1);
    const char* tn = sizeof(T) == 8 ? "f64" : "f32";
    const T c0 = (T)(1.0 / 3.0), c1 = (T)(1.0 / 6.0);
    auto report = [&](const char* name, float ms, double bytes_per_pt) {
        printf("L=%d %s %-34s %8.3f ms  %8.1f GB/s alg  %7.2f Gupd/s\n", level, tn, name, ms,
               pts * bytes_per_pt / (ms * 1e-3) / 1e9, pts / (ms * 1e-3) / 1e9);
        fflush(stdout);
    };
    if (!g_fused_only)
    {   // ceiling: same arrays, 2 reads + 1 write, grid-stride
        for (int blocks : {2048, 4096, 16384}) {
            float ms = tm.run([&] { hipLaunchKernelGGL(k_stream3<T>, dim3(blocks), dim3(256), 0, 0, u, b, tmp, elems / W); }, 3, iters);
            char nm[64]; snprintf(nm, sizeof nm, "stream 2R+1W (grid %d)", blocks);
            report(nm, ms, 3.0 * sizeof(T) * (double)elems / pts);
        }
    }
    for (int R : {4, 8, 16, 32, 64, 128}) {
        if (g_fused_only) break;
        const Launch g = make_launch(N, W, N - 1, R);
        T* src = u; T* dst = tmp;
        float ms = tm.run([&] {
            hipLaunchKernelGGL((k_jacobi<T>), dim3(g.blocks), dim3(kBlock), 0, 0, src, b, dst, N, pitch, 1, N, g.R, g.strips, g.chunks, c0, c1, N + 1);
            std::swap(src, dst);
        }, 3, iters);
        char nm[64]; snprintf(nm, sizeof nm, "jacobi A (62-lane) R=%d blocks=%d", R, g.blocks);
        report(nm, ms, 3.0 * sizeof(T));
    }
    for (int R : {8, 16, 32, 64}) {
        if (g_fused_only) break;
        Launch g = make_launch(N, W, N - 1, R);
        g.strips = (N / W + 63) / 64;
        long waves = (long)g.strips * g.chunks;
        g.blocks = (int)(((waves + 3) / 4 + 7) / 8 * 8);
        T* src = u; T* dst = tmp;
        float ms = tm.run([&] {
            hipLaunchKernelGGL((k_jacobi_b<T>), dim3(g.blocks), dim3(kBlock), 0, 0, src, b, dst, N, pitch, 1, N, g.R, g.strips, g.chunks, c0, c1);
            std::swap(src, dst);
        }, 3, iters);
        char nm[64]; snprintf(nm, sizeof nm, "jacobi B (64-lane+edge) R=%d", R);
        report(nm, ms, 3.0 * sizeof(T));
    }
    {
        const int strips = (N / W + kOutLanes - 1) / kOutLanes;
        const long waves = (long)strips * (N - 1);
        const int blocks = (int)(((waves + 3) / 4 + 7) / 8 * 8);
        T* src = u; T* dst = tmp;
        float ms = tm.run([&] {
            hipLaunchKernelGGL((k_jacobi_naive<T>), dim3(blocks), dim3(kBlock), 0, 0, src, b, dst, N, pitch, 1, N, strips, c0, c1);
            std::swap(src, dst);
        }, 3, iters);
        report("jacobi naive (row per wave)", ms, 3.0 * sizeof(T));
    }
    auto fused = [&](auto kc, int R) {
        constexpr int K = decltype(kc)::value;
        constexpr int OUT = fused_out_lanes<K, W>();
        Launch g = make_launch(N, W, N - 1, R);
        g.strips = (N / W + OUT - 1) / OUT;
        const long waves = (long)g.strips * g.chunks;
        g.blocks = (int)(((waves + 3) / 4 + 7) / 8 * 8);
        T* src = u; T* dst = tmp;
        float ms = tm.run([&] {
            hipLaunchKernelGGL((k_jacobi_fused<T, K, 0>), dim3(g.blocks), dim3(kBlock), 0, 0, src, b, dst, N, pitch, 1, N, g.R, g.strips, g.chunks, c0, c1, 0, N, 0, N + 1, 0);
            std::swap(src, dst);
        }, 3, iters);
        char nm[64]; snprintf(nm, sizeof nm, "jacobi fused K=%d R=%d (per sweep)", K, R);
        report(nm, ms / K, 3.0 * sizeof(T));
    };
    auto fused_gs = [&](auto kc, int R) {
        constexpr int K = decltype(kc)::value;       // levels = 2 x sweeps
        constexpr int OUT = fused_out_lanes<K, W>();
        Launch g = make_launch(N, W, N - 1, R);
        g.strips = (N / W + OUT - 1) / OUT;
        const long waves = (long)g.strips * g.chunks;
        g.blocks = (int)(((waves + 3) / 4 + 7) / 8 * 8);
        T* src = u; T* dst = tmp;
        float ms = tm.run([&] {
            hipLaunchKernelGGL((k_jacobi_fused<T, K, 1>), dim3(g.blocks), dim3(kBlock), 0, 0, src, b, dst, N, pitch, 1, N, g.R, g.strips, g.chunks, c0, c1, 0, N, 0, N + 1, 0);
            std::swap(src, dst);
        }, 3, iters);
        char nm[64]; snprintf(nm, sizeof nm, "rbgs fused sweeps=%d R=%d (per sweep)", K / 2, R);
        report(nm, ms / (K / 2), 3.0 * sizeof(T));
    };
    for (int R : {8, 16, 32, 64, 128}) {
        if (R * 4 > N) continue;
        fused_gs(std::integral_constant<int, 2>{}, R);
        fused_gs(std::integral_constant<int, 4>{}, R);
        fused_gs(std::integral_constant<int, 6>{}, R);
        fused_gs(std::integral_constant<int, 8>{}, R);
        fused_gs(std::integral_constant<int, 10>{}, R);
    }
    for (int R : {8, 16, 32}) {
        const Launch g = make_launch(N, W, N - 1, R);
        T* src = u; T* dst = tmp;
        float ms = tm.run([&] {
            hipLaunchKernelGGL((k_rbgs<T>), dim3(g.blocks), dim3(kBlock), 0, 0, src, b, dst, N, pitch, 1, N, g.R, g.strips, g.chunks, 0, 0, N);
            std::swap(src, dst);
        }, 3, iters);
        char nm[64]; snprintf(nm, sizeof nm, "rbgs one-pass k_rbgs R=%d", R);
        report(nm, ms, 3.0 * sizeof(T));
    }
    for (int R : {8, 16, 32, 64, 128}) {
        if (R * 4 > N) continue;
        fused(std::integral_constant<int, 2>{}, R);
        fused(std::integral_constant<int, 3>{}, R);
        fused(std::integral_constant<int, 4>{}, R);
        fused(std::integral_constant<int, 5>{}, R);
        fused(std::integral_constant<int, 6>{}, R);
        fused(std::integral_constant<int, 8>{}, R);
        fused(std::integral_constant<int, 10>{}, R);
    }
    if (g_fused_only) {
        CK(hipFree(u)); CK(hipFree(b)); CK(hipFree(tmp)); CK(hipFree(cb)); CK(hipFree(cu));
        return;
    }
    for (int R : {8, 16, 32, 64}) {
        const Launch g = make_launch(N, W, N - 1, R);
        T* src = u; T* dst = tmp;
        float ms = tm.run([&] {
            hipLaunchKernelGGL((k_rbgs<T>), dim3(g.blocks), dim3(kBlock), 0, 0, src, b, dst, N, pitch, 1, N, g.R, g.strips, g.chunks, 0, 0, N);
            std::swap(src, dst);
        }, 3, iters);
        char nm[64]; snprintf(nm, sizeof nm, "rbgs one-pass R=%d", R);
        report(nm, ms, 3.0 * sizeof(T));
    }
    for (int R : {8, 16, 32}) {
        const Launch g = make_launch(N, W, NC - 1, R);
        float ms = tm.run([&] {
            hipLaunchKernelGGL((k_restrict<T, true>), dim3(g.blocks), dim3(kBlock), 0, 0, u, b, cb, cu, N, pitch, cpitch, 1, NC, 0, g.R, g.strips, g.chunks, (T)0.25);
        }, 3, iters);
        char nm[64]; snprintf(nm, sizeof nm, "residual+restrict fused Rc=%d", R);
        report(nm, ms, 2.0 * sizeof(T) + 0.5 * sizeof(T));
    }
    for (int R : {8, 16, 32}) {
        const Launch g = make_launch(N, W, N - 1, R);
        float ms = tm.run([&] {
            hipLaunchKernelGGL((k_prolong<T, true>), dim3(g.blocks), dim3(kBlock), 0, 0, u, cu, N, pitch, cpitch, 1, N, 0, g.R, g.strips, g.chunks);
        }, 3, iters);
        char nm[64]; snprintf(nm, sizeof nm, "prolong+add R=%d", R);
        report(nm, ms, 2.0 * sizeof(T) + 0.25 * sizeof(T));
        // undo growth of u is unnecessary: values only drift linearly
    }
    {
        const Launch g = make_launch(N, W, N - 1, 0);
        double* partial; double* sum;
        CK(hipMalloc(&partial, (g.blocks + 8) * sizeof(double))); CK(hipMalloc(&sum, 8));
        float ms = tm.run([&] {
            hipLaunchKernelGGL((k_residual<T, 1>), dim3(g.blocks), dim3(kBlock), 0, 0, u, b, (void*)nullptr, 0L, partial, 1.0, N, pitch, 1, N, g.R, g.strips, g.chunks, N + 1);
            hipLaunchKernelGGL(k_reduce_partials, dim3(1), dim3(kReduceThreads), 0, 0, partial, g.blocks, sum);
        }, 3, iters);
        report("residual norm (2 launches)", ms, 2.0 * sizeof(T));
        CK(hipFree(partial)); CK(hipFree(sum));
    }
    CK(hipFree(u)); CK(hipFree(b)); CK(hipFree(tmp)); CK(hipFree(cb)); CK(hipFree(cu));
}

int main(int argc, char** argv)
{
    const int level = argc > 1 ? atoi(argv[1]) : 13;
    const int iters = argc > 2 ? atoi(argv[2]) : 20;
    const char* which = argc > 3 ? argv[3] : "both";
    g_fused_only = argc > 4 && argv[4][0] == 'f';
    if (level < 6 || level > 14) { printf("level must be 6..14\n"); return 1; }
    hipDeviceProp_t p;
    CK(hipGetDeviceProperties(&p, 0));
    printf("device: %s, %d CUs, %.1f GiB\n", p.name, p.multiProcessorCount, p.totalGlobalMem / 1073741824.0);
    if (which[0] == 'b' || which[0] == 'd') run_level<double>(level, iters);
    if (which[0] == 'b' || which[0] == 'f') run_level<float>(level, iters);
    return 0;
}
