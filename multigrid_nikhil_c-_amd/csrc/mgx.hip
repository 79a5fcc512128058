// mgx.hip — libmgx: solver handle, V-cycle / FMG schedules and the C-ABI
// (include/mgx.h) over the gfx950 kernels of mgx_kernels.hpp and their launch wrappers /
// pass planner of mgx_launch.hpp.
//
// Reference map (PS = Poissons_SYCL.cpp, MF = Multigrid_functions.cpp):
//   Solver / Level      PS:24-33 matrix_elements_for_jacobi + jacobi_matrices[],
//                       MF:16-26 ProblemVar   (matrix-free: arrays, no CSR)
//   Solver::vcycle      PS:575-627 vcyclemultigrid / MF:132-173
//   Solver::fmg         PS:629-650 fullmultigrid   / MF:175-191
//   mgx_solve           PS:727 (main's call) / MF:193-197 multigrid_solver
// There is no CPU fallback anywhere in this file.

#include "../../include/mgx.h"
#include "mgx_bottom.hpp"
#include "mgx_kernels.hpp"
#include "mgx_launch.hpp"
#include "mgx_var.hpp"
#include "mgx_dist_plan.hpp"

#include <chrono>
#include <unistd.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

namespace {

using namespace mgx;

thread_local std::string g_create_error;

// MGX_LOG_RUNTIME_LIBS=1: print the mapped ROCm runtime libraries when a handle is created (mgx_runtime_libs)
void log_runtime_libs(const char* where)
{
    if (env_int("MGX_LOG_RUNTIME_LIBS", 0) == 0) return;
    char buf[8192];
    if (mgx_runtime_libs(buf, sizeof buf) >= 0)
        std::fprintf(stderr, "[mgx] runtime libraries mapped at %s (pid %d):\n%s", where, (int)getpid(), buf);
}

struct Level {
    int L = 0, N = 0, rows = 0;
    long pitch = 0;
    size_t bytes = 0;
    bool f64 = true;
    void *u = nullptr, *b = nullptr, *tmp = nullptr, *r = nullptr;
    // MGX_OPERATOR_STENCIL5 (mgx_var.hpp): A = (c, n, s, w, e)  [ProblemVar::A_sp_dict, MF:19] and its Jacobi
    // splitting (D_inv, R_n, R_s, R_w, R_e)  [A_jacobi_sp_dict, MF:20, 28-32]
    void* coef[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
    void* jac[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
    bool stencil_set = false;
    size_t esize() const { return f64 ? 8 : 4; }
};

struct EventPair { hipEvent_t a, b; int cls; long long launches; long long sweeps; hipEvent_t a_use; };   // a_use: the event the span starts at (a, or the previous span's b)

} // namespace

struct mgx_solver {
    mgx_config cfg{};
    hipStream_t stream = nullptr;
    std::vector<Level> lv;          // working hierarchy, index = level (only [coarsest..finest] valid)
    Level fine64;                   // MIXED only: double u, b (and r) on the finest level
    bool mixed = false;
    bool work_f64 = true;           // element type of the working hierarchy
    BottomDST bottom;
    double* partial = nullptr;      // per-block partial sums
    long partial_cap = 0;
    double* sum_dev = nullptr;      // reduced sum (device)
    double* sum_host = nullptr;     // pinned
    std::string err;
    int rows_per_chunk = 0;         // 0 = auto (MGX_ROWS env overrides)
    FuseCfg fuse{10, 0, 256, 10, 10, 1024, 10, 10};   // temporal fusion knobs (MGX_FUSE, MGX_FUSE_ROWS, MGX_FUSE_MIN_N, MGX_FOLD_KMAX[_BIG])
    // profiling
    std::vector<EventPair> ev_used, ev_free;
    double prof_ms[MGX_PROF_COUNT] = {0};
    long long prof_launches[MGX_PROF_COUNT] = {0};
    long long prof_sweeps[MGX_PROF_COUNT] = {0};
    int last_smooth_launches = 0;   // launches made by the most recent smoothing block
    int fold = 1;                   // fold transfers / norm into smoother passes (MGX_FOLD)
    int use_zero_in = 1;            // let first passes synthesise a known-zero iterate (MGX_ZERO_IN)
    int zero_in_level = -1;         // level whose U is known to be all zero and has NOT been zero-filled
    bool want_norm = false;         // the top-level post-smoothing should also produce ||r||^2 partials
    int norm_blocks_ready = 0;      // > 0: partial[] holds that many sums of r^2 for the current U
    double fine_updates = 0.0;
    // hipGraph replay of "one V-cycle + residual norm" in mgx_solve (profiling off only)
    struct CycleGraph {
        std::vector<void*> before, after;   // u / tmp of every level before and after the cycle
        hipGraphExec_t exec = nullptr;
        double fine_updates = 0.0;
    };
    std::vector<CycleGraph> graphs;
    int use_graph = 1;              // MGX_GRAPH
    bool prof_mute = false;         // inside a stream capture: no events (they carry no time stamps there)
    // cfg.profile = 2: the spans of one cycle follow one another with nothing enqueued in between, so the end event of a span is
    // the start of the next (an event record costs ~4 us of GPU time: 8 per cycle were 2.5 % of it, 5 are 1.6 %)
    bool prof_chain = false;
    hipEvent_t chain_ev = nullptr;
    int mixed_fuse = 1;             // mixed precision: u += s e and the residual in one pass (MGX_MIXED_FUSE)
    // general per-level operators: dense inverse of the coarsest one (MF:18 coarsest_level_matrix, MF:63-72)
    double *var_M = nullptr, *var_inv = nullptr, *var_pm = nullptr, *var_pi = nullptr;
    bool var = false;               // cfg.op == MGX_OPERATOR_STENCIL5
    struct mgx_dist* dist = nullptr; // multi-GPU handle (cfg.n_gpus > 1 / mgx_create_rank): mgx_dist.hpp; no levels of its own

    int fail(int code, const std::string& m) { err = m; return code; }
};

namespace {

#define HIPCHK(h, expr)                                                                  \
    do {                                                                                 \
        hipError_t e__ = (expr);                                                         \
        if (e__ != hipSuccess)                                                           \
            return (h)->fail(MGX_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e__)); \
    } while (0)

// ---- profiling scopes ---------------------------------------------------------
struct Prof {
    mgx_solver* s; int idx = -1;
    Prof(mgx_solver* s_, int cls, long long launches) : s(s_)
    {
        static const bool mute_all = env_int("MGX_PROF_MUTE", 0) != 0;      // experiment: the split submission of cfg.profile = 2 without its events
        if (!s->cfg.profile || s->prof_mute || mute_all) return;
        EventPair p;
        if (!s->ev_free.empty()) { p = s->ev_free.back(); s->ev_free.pop_back(); }
        else {
            // timing only: no system-scope fence when the event completes (MGX_PROF_EVENT_FLAGS=0: plain events)
            static const unsigned flags = env_int("MGX_PROF_EVENT_FLAGS", 1) ? hipEventDisableSystemFence : hipEventDefault;
            if (hipEventCreateWithFlags(&p.a, flags) != hipSuccess || hipEventCreateWithFlags(&p.b, flags) != hipSuccess) return;
        }
        p.cls = cls; p.launches = launches; p.sweeps = 0;
        p.a_use = p.a;
        if (s->prof_chain && s->chain_ev) p.a_use = s->chain_ev;
        else (void)hipEventRecord(p.a, s->stream);
        s->ev_used.push_back(p);
        idx = (int)s->ev_used.size() - 1;
    }
    void set(long long launches, long long sweeps)
    {
        if (idx >= 0) { s->ev_used[idx].launches = launches; s->ev_used[idx].sweeps = sweeps; }
    }
    ~Prof()
    {
        if (idx < 0) return;
        (void)hipEventRecord(s->ev_used[idx].b, s->stream);
        if (s->prof_chain) s->chain_ev = s->ev_used[idx].b;
    }
};

int prof_collect(mgx_solver* s)
{
    if (s->ev_used.empty()) return MGX_OK;
    HIPCHK(s, hipStreamSynchronize(s->stream));
    for (auto& p : s->ev_used) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, p.a_use, p.b) == hipSuccess) {
            s->prof_ms[p.cls] += ms;
            s->prof_launches[p.cls] += p.launches;
            s->prof_sweeps[p.cls] += p.sweeps;
        }
        s->ev_free.push_back(p);
    }
    s->ev_used.clear();
    s->chain_ev = nullptr;
    return MGX_OK;
}

// ---- device-side fills -------------------------------------------------------------
template <typename T>
__global__ void k_fill_rhs(T* b, int N, long pitch, int kind, double f)
{
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    const int r = blockIdx.y;
    if (c > N || r > N) return;
    const double h = 1.0 / (double)N;
    double v = 0.0;
    if (r >= 1 && r < N && c >= 1 && c < N) {
        if (kind == 0) v = f * h * h;                                   // PS:283-335 (sign: D1)
        else v = h * h * 8.0 * 9.869604401089358 * sinpi(2.0 * c * h) * sinpi(2.0 * r * h);
    }
    b[(long)r * pitch + c] = (T)v;
}

__device__ inline uint64_t splitmix64(uint64_t x)
{
    x += 0x9E3779B97F4A7C15ULL;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ULL;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBULL;
    return x ^ (x >> 31);
}

template <typename T>
__global__ void k_fill_random(T* u, int N, long pitch, uint64_t seed)
{
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    const int r = blockIdx.y;
    if (c > N || r > N) return;
    double v = 0.0;
    if (r >= 1 && r < N && c >= 1 && c < N) {
        const uint64_t idx = (uint64_t)(r - 1) * (uint64_t)(N - 1) + (uint64_t)(c - 1);  // PS:227 numbering
        const uint64_t x = splitmix64(seed ^ splitmix64(idx));
        v = (double)(x >> 11) * (1.0 / 4503599627370496.0) - 1.0;
    }
    u[(long)r * pitch + c] = (T)v;
}

// ---- level helpers ------------------------------------------------------------------
int alloc_level(mgx_solver* s, Level& l, int level, bool f64)
{
    l.L = level;
    l.N = 1 << level;
    l.rows = l.N + 1;
    l.f64 = f64;
    l.pitch = level_pitch(level, f64 ? MGX_DTYPE_F64 : MGX_DTYPE_F32);
    l.bytes = (size_t)l.rows * (size_t)l.pitch * l.esize();
    for (void** p : {&l.u, &l.b, &l.tmp}) {
        if (hipMalloc(p, l.bytes) != hipSuccess) return s->fail(MGX_ERR_ALLOC, "hipMalloc failed for level arrays");
        HIPCHK(s, hipMemsetAsync(*p, 0, l.bytes, s->stream));
    }
    return MGX_OK;
}

void free_level(Level& l)
{
    for (void** p : {&l.u, &l.b, &l.tmp, &l.r, &l.coef[0], &l.coef[1], &l.coef[2], &l.coef[3], &l.coef[4],
                     &l.jac[0], &l.jac[1], &l.jac[2], &l.jac[3], &l.jac[4]}) {
        if (*p) (void)hipFree(*p);
        *p = nullptr;
    }
}

int ensure_r(mgx_solver* s, Level& l)
{
    if (l.r) return MGX_OK;
    if (hipMalloc(&l.r, l.bytes) != hipSuccess) return s->fail(MGX_ERR_ALLOC, "hipMalloc failed for residual array");
    HIPCHK(s, hipMemsetAsync(l.r, 0, l.bytes, s->stream));
    return MGX_OK;
}

bool level_ok(const mgx_solver* s, int level)
{
    return level >= s->cfg.coarsest_level && level <= s->cfg.finest_level;
}

// interior (n x n, reference layout) <-> padded grid
int copy_in(mgx_solver* s, Level& l, void* dst_grid, const void* src, size_t count)
{
    const size_t n = (size_t)l.N - 1;
    if (count != n * n) return s->fail(MGX_ERR_INVALID, "vector length must be n*n with n = 2^level - 1");
    const size_t es = l.esize();
    char* d = reinterpret_cast<char*>(dst_grid) + ((size_t)l.pitch + 1) * es;   // (row 1, col 1)
    HIPCHK(s, hipMemcpy2DAsync(d, (size_t)l.pitch * es, src, n * es, n * es, n, hipMemcpyHostToDevice, s->stream));
    HIPCHK(s, hipStreamSynchronize(s->stream));
    return MGX_OK;
}

int copy_out(mgx_solver* s, Level& l, const void* src_grid, void* dst, size_t count)
{
    const size_t n = (size_t)l.N - 1;
    if (count != n * n) return s->fail(MGX_ERR_INVALID, "vector length must be n*n with n = 2^level - 1");
    const size_t es = l.esize();
    const char* p = reinterpret_cast<const char*>(src_grid) + ((size_t)l.pitch + 1) * es;
    HIPCHK(s, hipMemcpy2DAsync(dst, n * es, p, (size_t)l.pitch * es, n * es, n, hipMemcpyDeviceToHost, s->stream));
    HIPCHK(s, hipStreamSynchronize(s->stream));
    return MGX_OK;
}

// ---- operators on the working hierarchy -----------------------------------------------
inline bool tile_level(const mgx_solver* s, const Level& l) { return s->fuse.tile_max_n > 0 && l.N <= s->fuse.tile_max_n; }

template <typename T>
void smooth_t(mgx_solver* s, Level& l, int mu)
{
    int parity = 0, launches = 0;
    if (tile_level(s, l)) {
        FoldArgs fa;
        const int rc = s->cfg.smoother == MGX_SMOOTHER_RBGS
            ? smooth_tiled<T, 1, 0>((T*)l.u, (const T*)l.b, (T*)l.tmp, l.N, l.pitch, mu, s->cfg.omega, s->fuse.tile_k, fa, false, 0, false, s->stream, &launches)
            : (s->fuse.arith
               ? smooth_tiled<T, 0, 1>((T*)l.u, (const T*)l.b, (T*)l.tmp, l.N, l.pitch, mu, s->cfg.omega, s->fuse.tile_k, fa, false, 0, false, s->stream, &launches)
               : smooth_tiled<T, 0, 0>((T*)l.u, (const T*)l.b, (T*)l.tmp, l.N, l.pitch, mu, s->cfg.omega, s->fuse.tile_k, fa, false, 0, false, s->stream, &launches));
        if (rc >= 0) {
            s->last_smooth_launches = launches;
            if (launches & 1) std::swap(l.u, l.tmp);
            return;
        }
    }
    (void)smooth_block<T>(s->cfg.smoother, (T*)l.u, (const T*)l.b, (T*)l.tmp, l.N, l.pitch, l.rows, 1, l.N, mu,
                          s->cfg.omega, false, 1, l.N, 0, s->rows_per_chunk, s->fuse, s->stream, &parity, &launches);
    s->last_smooth_launches = launches;
    if (parity) std::swap(l.u, l.tmp);
}

// ---- general per-level operators (cfg.op = MGX_OPERATOR_STENCIL5; kernels in mgx_var.hpp) ---------------
int var_alloc_level(mgx_solver* s, Level& l)
{
    for (void** p : {&l.coef[0], &l.coef[1], &l.coef[2], &l.coef[3], &l.coef[4], &l.jac[0], &l.jac[1], &l.jac[2], &l.jac[3], &l.jac[4]}) {
        if (hipMalloc(p, l.bytes) != hipSuccess) return s->fail(MGX_ERR_ALLOC, "hipMalloc failed for the operator's coefficient arrays");
        HIPCHK(s, hipMemsetAsync(*p, 0, l.bytes, s->stream));
    }
    return MGX_OK;
}

// the level's operator has been written into l.coef[]: build {D_inv, R_omega} (MF:28-32) and, on the coarsest
// level with an exact bottom solve, the dense inverse (MF:63-72)
template <typename T>
int var_build_t(mgx_solver* s, Level& l)
{
    const dim3 blk(256), grd((l.N + 1 + 255) / 256, l.N + 1);
    hipLaunchKernelGGL((k_var_build_jacobi<T>), grd, blk, 0, s->stream, (const T*)l.coef[0], (const T*)l.coef[1], (const T*)l.coef[2],
                       (const T*)l.coef[3], (const T*)l.coef[4], (T*)l.jac[0], (T*)l.jac[1], (T*)l.jac[2], (T*)l.jac[3], (T*)l.jac[4],
                       l.N, l.pitch, (T)s->cfg.omega);
    if (l.L == s->cfg.coarsest_level && s->cfg.bottom == MGX_BOTTOM_EXACT) {
        const int n = l.N - 1, NN = n * n;
        hipLaunchKernelGGL((k_var_dense_fill<T>), dim3((NN + 255) / 256, NN), dim3(256), 0, s->stream, s->var_M, s->var_inv,
                           (const T*)l.coef[0], (const T*)l.coef[1], (const T*)l.coef[2], (const T*)l.coef[3], (const T*)l.coef[4], n, l.pitch);
        for (int k = 0; k < NN; ++k) {
            hipLaunchKernelGGL(k_gj_prow, dim3((NN + 255) / 256), dim3(256), 0, s->stream, s->var_M, s->var_inv, s->var_pm, s->var_pi, NN, k);
            hipLaunchKernelGGL(k_gj_elim, dim3(NN), dim3(256), 0, s->stream, s->var_M, s->var_inv, s->var_pm, s->var_pi, NN, k);
        }
    }
    HIPCHK(s, hipGetLastError());
    HIPCHK(s, hipStreamSynchronize(s->stream));
    l.stencil_set = true;
    for (auto& g : s->graphs) if (g.exec) (void)hipGraphExecDestroy(g.exec);
    s->graphs.clear();       // (the kernels' coefficient pointers are unchanged, but a new operator is a new problem: recapture)
    return MGX_OK;
}
int var_build(mgx_solver* s, Level& l) { return l.f64 ? var_build_t<double>(s, l) : var_build_t<float>(s, l); }

// every level's operator must have been given before a schedule or operator runs
int var_ready(mgx_solver* s, int lo, int hi)
{
    if (!s->var) return MGX_OK;
    for (int l = lo; l <= hi; ++l)
        if (!s->lv[l].stencil_set)
            return s->fail(MGX_ERR_STATE, "operator of level " + std::to_string(l) + " not set (mgx_set_stencil / mgx_set_coefficient)");
    return MGX_OK;
}

// MF:75-96: mu sweeps, one launch each, u <-> tmp
template <typename T>
void smooth_var_t(mgx_solver* s, Level& l, int mu)
{
    const T om = (T)s->cfg.omega;
    const T rc = (T)(1.0 - (double)om);
    const Launch g = make_launch(l.N, VecOf<T>::W, l.N - 1, 1);
    for (int i = 0; i < mu; ++i) {
        hipLaunchKernelGGL((k_jacobi_var<T>), dim3(g.blocks), dim3(kBlock), 0, s->stream, (const T*)l.u, (const T*)l.b, (T*)l.tmp,
                           (const T*)l.jac[0], (const T*)l.jac[1], (const T*)l.jac[2], (const T*)l.jac[3], (const T*)l.jac[4],
                           l.N, l.pitch, 1, l.N, g.strips, rc, om, l.rows);
        std::swap(l.u, l.tmp);
    }
    s->last_smooth_launches = mu;
}

// MF:150-153: r = b - A u into `out` (MODE 0) or sum r^2 -> sum_dev (MODE 1)
template <typename T, int MODE>
void residual_var_t(mgx_solver* s, const Level& l, const void* u, const void* b, void* out)
{
    const Launch g = make_launch(l.N, VecOf<T>::W, l.N - 1, 1);
    hipLaunchKernelGGL((k_residual_var<T, MODE>), dim3(g.blocks), dim3(kBlock), 0, s->stream, (const T*)u, (const T*)b, (T*)out,
                       s->partial, (const T*)l.coef[0], (const T*)l.coef[1], (const T*)l.coef[2], (const T*)l.coef[3],
                       (const T*)l.coef[4], l.N, l.pitch, 1, l.N, g.strips, l.rows);
    if (MODE == 1) hipLaunchKernelGGL(k_reduce_partials, dim3(1), dim3(kReduceThreads), 0, s->stream, s->partial, g.blocks, s->sum_dev);
}

// ---- planning the passes of a folded smoothing block ------------------------------------------
// Pass costs relative to a pass of up to 5 levels (HBM-bound: ~385 us at 8192^2 in double whatever
// its depth), from bench.py runs with explicit plans (MGX_PLAN_PRE / MGX_PLAN_POST) against each other
// inside one GPU call.  Round 2, double, after the deep passes got their rhs window in LDS and
// branch-free interior bodies (rows really in flight): 6 levels 1.05, 8 levels 1.12, 10 levels 1.46
// (0.56 ms) - a 10-level pass now costs less than two 5-level ones, so V(10,10) is planned as ONE
// pass per block ([10]: 1.93 ms per cycle against 2.35 as [5,5], 2.28 as [8,2], 2.35 as [6,4]).
// float (no LDS variants, packed arithmetic): 6: 1.34, 8: 1.53; red-black Gauss-Seidel (levels =
// 2 x sweeps) 6: 1.1, 8: 1.3, 10: 2.4.  kcap: MGX_FOLD_KMAX / _BIG / _NOPOST / _GS still cap the depth.
inline int fold_kcap(const FuseCfg& f, int smoother, int N, int post, bool f64)
{
    int k = N >= 8192 ? f.fold_kmax_big : f.fold_kmax;
    if (post == 0) k = std::min(k, f.fold_kmax_nopost);
    if (smoother == MGX_SMOOTHER_RBGS) k = std::min(k, env_int("MGX_FOLD_KMAX_GS", 10));
    return k;
}

inline double fold_pass_cost(int K, int smoother, int N, int post, bool f64, int arith)
{
    const bool rbgs = (smoother == MGX_SMOOTHER_RBGS);
    if (!cycle_k_supported(K, rbgs, f64, post, false, arith)) return -1.0;
    (void)N;
    if (rbgs) return K <= 4 ? 1.0 : (K == 6 ? 1.1 : (K == 8 ? 1.3 : 2.4));
    if (K <= 5) return 1.0;
    if (!f64) return K == 6 ? 1.34 : (K == 8 ? 1.53 : 1.75);      // (10 levels: rhs window in LDS since round 3)
    if (K == 6) return 1.05;
    if (K == 8) return 1.12;
    return 1.46;
}

// parts[] = sweeps per pass, deepest first (the last pass carries the residual stage, which is
// what gets expensive with depth; a leading single sweep could not synthesise a zero input)
inline int plan_folded(const FuseCfg& f, int smoother, int N, int mu, int post, bool f64, int* parts, bool pre = false)
{
    const int per = (smoother == MGX_SMOOTHER_RBGS) ? 2 : 1;
    const int smax = std::max(1, fold_kcap(f, smoother, N, post, f64) / per);
    std::vector<double> best(mu + 1, 1e300);
    std::vector<int> pick(mu + 1, 1);
    best[0] = 0.0;
    for (int m = 1; m <= mu; ++m)
        for (int k = 1; k <= std::min(m, smax); ++k) {
            const double c = fold_pass_cost(per * k, smoother, N, post, f64, f.arith);
            if (c < 0.0) continue;
            // a block done in ONE pass carries the correction AND the residual stage: some depths exist for either only
            if (k == mu && m == mu && !cycle_k_supported(per * k, smoother == MGX_SMOOTHER_RBGS, f64, post, pre, f.arith)) continue;
            const double t = best[m - k] + c + 1e-3;      // equal sums: fewer passes
            if (t < best[m] - 1e-12) { best[m] = t; pick[m] = k; }
        }
    int n = 0;
    for (int m = mu; m > 0; m -= pick[m]) parts[n++] = pick[m];
    std::sort(parts, parts + n, [](int a, int b) { return a > b; });
    return n;
}

// The passes (sweeps per pass) of a folded smoothing block: pre-smoothing = (pre false, post 1),
// post-smoothing = (pre true, post 0 or 2).  An explicit plan wins when it fits; otherwise the
// DP over the measured rates, capped at the folded kernels' depth.
int fold_plan(const mgx_solver* s, const Level& l, int mu, bool pre, int post, int* parts)
{
    const bool rbgs = (s->cfg.smoother == MGX_SMOOTHER_RBGS);
    const FuseCfg& f = s->fuse;
    const int* forced = nullptr;
    int nf = 0;
    if (!pre && post == 1) { forced = f.plan_pre; nf = f.n_pre; }
    else if (pre) { forced = f.plan_post; nf = f.n_post; }
    if (nf > 0 && l.N >= f.plan_min_n) {
        const int per = rbgs ? 2 : 1;
        int sum = 0;
        bool ok = true;
        for (int i = 0; i < nf; ++i) {
            sum += forced[i];
            const int K = per * forced[i];
            const bool folded = (i == 0 && pre) || (i == nf - 1 && post != 0);
            const int q = (i == nf - 1) ? post : 0;
            ok = ok && K <= 10 && (folded ? cycle_k_supported(K, rbgs, l.f64, q, pre && i == 0, f.arith) : (K != 7 && K != 9 && (!rbgs || K % 2 == 0)));
        }
        if (ok && sum == mu) {
            for (int i = 0; i < nf; ++i) parts[i] = forced[i];
            return nf;
        }
    }
    return plan_folded(f, s->cfg.smoother, l.N, mu, post, l.f64, parts, pre);
}

// mu Jacobi sweeps on a whole level with the prolongation+correction applied while
// loading (pre_e: coarse correction, may be null) and/or the residual restriction
// (post = 1) or the residual norm (post = 2) produced by the last pass.
// Returns false when this level / configuration is not eligible (caller then
// uses the stand-alone kernels); on success *norm_blocks = partial sums written.

template <typename T, int SM, int AR>
bool smooth_folded_t(mgx_solver* s, Level& l, int mu, const Level* coarse, bool pre, int post, int* launches,
                     int* norm_blocks, bool zero_in)
{
    constexpr bool rbgs = (SM == 1);
    constexpr int per = rbgs ? 2 : 1;
    FoldArgs fa;
    fa.restrict_mode = s->cfg.restrict_mode;
    fa.partial = s->partial;
    if (coarse) {
        fa.cpitch = coarse->pitch; fa.coarse_e = coarse->u; fa.coarse_b = coarse->b;
        // PS:613: zero the coarse guess here unless its first pre-smoothing pass synthesises it
        fa.coarse_zero = (post == 1 && s->zero_in_level == coarse->L) ? nullptr : coarse->u;
    }
    if (tile_level(s, l)) {
        int flips = 0;
        const int nb = smooth_tiled<T, SM, AR>((T*)l.u, (const T*)l.b, (T*)l.tmp, l.N, l.pitch, mu, s->cfg.omega,
                                           s->fuse.tile_k, fa, pre, post, zero_in, s->stream, &flips);
        if (nb < 0) return false;
        if (flips & 1) std::swap(l.u, l.tmp);
        if (post == 2) *norm_blocks = nb;
        *launches = flips;
        return true;
    }
    int parts[64];
    const int np = fold_plan(s, l, mu, pre, post, parts);
    const T om = (T)s->cfg.omega;
    const T c0 = (T)(1.0 - (double)om);
    const T c1 = (T)((double)om / 4.0);
    T* src = (T*)l.u; T* dst = (T*)l.tmp;
    const T* b = (const T*)l.b;
    for (int p = 0; p < np; ++p) {
        const bool first = (p == 0), last = (p == np - 1);
        const bool P = pre && first;
        const int Q = last ? post : 0;
        const int K = per * parts[p];
        const int R = fuse_rows(s->fuse, l.N, K, sizeof(T) == 8);
        const int Rc = fuse_rows_auto(s->fuse, l.N, K, sizeof(T) == 8) ? -R : R;      // folded passes: sized by the launcher
        fa.zero_in = (first && zero_in) ? 1 : 0;
        int blocks = 0;
        if (P && Q == 2) blocks = launch_cycle<T, 1, 2, SM, AR>(K, src, b, dst, fa, l.N, l.pitch, c0, c1, Rc, s->stream);
        else if (P) blocks = launch_cycle<T, 1, 0, SM, AR>(K, src, b, dst, fa, l.N, l.pitch, c0, c1, Rc, s->stream);
        else if (Q == 1) blocks = launch_cycle<T, 0, 1, SM, AR>(K, src, b, dst, fa, l.N, l.pitch, c0, c1, Rc, s->stream);
        else if (Q == 2) blocks = launch_cycle<T, 0, 2, SM, AR>(K, src, b, dst, fa, l.N, l.pitch, c0, c1, Rc, s->stream);
        else if (!rbgs && K == 1) (void)launch_jacobi<T>(src, b, dst, l.N, l.pitch, 1, l.N, s->cfg.omega, s->rows_per_chunk, s->stream, l.rows, AR);
        else (void)launch_fused<T, SM, AR>(K, src, b, dst, l.N, l.pitch, 1, l.N, c0, c1, 0, l.N, 0, R, s->stream, l.rows, fa.zero_in);
        if (Q == 2) *norm_blocks = blocks;
        std::swap(src, dst);
    }
    if (np & 1) std::swap(l.u, l.tmp);
    *launches = np;
    return true;
}

// pre-check made before any launch (so a `false` never leaves a half-done block)
bool fold_eligible(const mgx_solver* s, const Level& l, int mu, bool pre = false, int post = 1)
{
    if (!s->fold || mu < 1 || mu > 64) return false;
    // general operators have no fused / folded kernels; the folded restriction is full weighting
    if (s->var || (post == 1 && s->cfg.restrict_mode >= MGX_RESTRICT_INJECT)) return false;
    if (tile_level(s, l)) return true;
    if (l.N < s->fuse.min_n) return false;
    const bool rbgs = (s->cfg.smoother == MGX_SMOOTHER_RBGS);
    const int per = rbgs ? 2 : 1;
    if (s->fuse.kmax < per) return false;
    int parts[64];
    const int np = fold_plan(s, l, mu, pre, post, parts);
    for (int p = 0; p < np; ++p)
        if (!cycle_k_supported(per * parts[p], rbgs, l.f64, p == np - 1 ? post : 0, pre && p == 0, s->fuse.arith)) return false;
    // the norm partials of the folded pass must fit the reduction buffer
    return true;
}

// May the pre-smoothing of `level` start from an implicit zero iterate (the producer then
// skips the zero fill)?  Only when its first pass is a kernel that honours zero_in: a folded
// or fused pass, not a stand-alone single sweep.
bool zero_in_ok(const mgx_solver* s, int level)
{
    if (!s->use_zero_in || level <= s->cfg.coarsest_level) return false;
    const Level& l = s->lv[level];
    const int mu = s->cfg.mu1;
    if (!fold_eligible(s, l, mu)) return false;
    if (tile_level(s, l)) return true;
    const bool rbgs = (s->cfg.smoother == MGX_SMOOTHER_RBGS);
    int parts[64];
    const int np = fold_plan(s, l, mu, false, 1, parts);
    return !(np >= 2 && !rbgs && parts[0] == 1);      // a leading plain single Jacobi sweep reads its input
}

bool smooth_folded(mgx_solver* s, int level, int mu, bool pre, int post, bool zero_in = false)
{
    Level& l = s->lv[level];
    if (!fold_eligible(s, l, mu, pre, post)) return false;
    const Level* coarse = (pre || post == 1) ? &s->lv[level - 1] : nullptr;
    const bool fine = (level == s->cfg.finest_level);
    const bool rbgs = (s->cfg.smoother == MGX_SMOOTHER_RBGS);
    const bool fma = s->fuse.arith != 0;
    // float: the 10-level pass with BOTH the correction stage and the norm stage does not fit its registers (mgx_launch.hpp,
    // cycle_k_supported) and a block of 10 would run as two passes of 5 (8192^2: 223 + 190 us); one 10-level pass without
    // the norm stage and the stand-alone norm kernel are 216 + ~110 us.  (The same for the restriction stage of the
    // separately rounded mode - 10 levels + the stand-alone residual / restriction instead of 8 + 2 - was measured and is
    // worse: mixed cycle 1.54 -> 1.72 ms, the stand-alone transfer alone is 0.15 ms.)
    static const bool f32_norm_apart = env_int("MGX_F32_NORM_APART", 1) != 0;
    if (!l.f64 && !rbgs && mu == 10 && l.N > s->fuse.tile_max_n && f32_norm_apart && post == 2 && pre &&
        cycle_k_supported(mu, false, false, 0, pre, s->fuse.arith) && !cycle_k_supported(mu, false, false, post, pre, s->fuse.arith))
        post = 0;
    {
        Prof p(s, fine ? MGX_PROF_SMOOTH_FINE : MGX_PROF_COARSE, mu);
        int launches = 0, nb = 0;
        bool ok;
        if (l.f64) ok = rbgs ? smooth_folded_t<double, 1, 0>(s, l, mu, coarse, pre, post, &launches, &nb, zero_in)
                             : (fma ? smooth_folded_t<double, 0, 1>(s, l, mu, coarse, pre, post, &launches, &nb, zero_in)
                                    : smooth_folded_t<double, 0, 0>(s, l, mu, coarse, pre, post, &launches, &nb, zero_in));
        else ok = rbgs ? smooth_folded_t<float, 1, 0>(s, l, mu, coarse, pre, post, &launches, &nb, zero_in)
                       : (fma ? smooth_folded_t<float, 0, 1>(s, l, mu, coarse, pre, post, &launches, &nb, zero_in)
                              : smooth_folded_t<float, 0, 0>(s, l, mu, coarse, pre, post, &launches, &nb, zero_in));
        if (!ok) return false;
        p.set(launches, mu);
        if (post == 2) s->norm_blocks_ready = nb;
    }
    // (the norm stage left out: mgx_solve's residual_norm_grid finds norm_blocks_ready == 0 and runs the norm kernel)
    if (fine) s->fine_updates += (double)mu * (double)(l.N - 1) * (double)(l.N - 1);
    return true;
}

void smooth(mgx_solver* s, int level, int mu)
{
    if (mu <= 0) return;
    Level& l = s->lv[level];
    const bool fine = (level == s->cfg.finest_level);
    Prof p(s, fine ? MGX_PROF_SMOOTH_FINE : MGX_PROF_COARSE, mu);
    if (s->var) { if (l.f64) smooth_var_t<double>(s, l, mu); else smooth_var_t<float>(s, l, mu); }       // MF:75-96
    else if (l.f64) smooth_t<double>(s, l, mu); else smooth_t<float>(s, l, mu);
    p.set(s->last_smooth_launches, mu);
    if (fine) s->fine_updates += (double)mu * (double)(l.N - 1) * (double)(l.N - 1);
}

// B[level-1] = R (B - A U)[level]  (fused = true)  or  R B[level]  (fused = false)
void restrict_level(mgx_solver* s, int level, bool fused, bool zero_guess)
{
    Level& f = s->lv[level];
    Level& c = s->lv[level - 1];
    const bool fine = (level == s->cfg.finest_level);
    Prof p(s, fine ? MGX_PROF_RESTRICT_FINE : MGX_PROF_COARSE, 1);
    const int rpc = s->rows_per_chunk;
    const int mode = s->cfg.restrict_mode;
    if (s->var || mode >= MGX_RESTRICT_INJECT) {
        // general operator and / or injection (MF:122-130): the residual is formed first (MF:150-153; f.r was
        // allocated with the handle), then restricted by full weighting (PS:531-546) or injected
        const void* src = f.b;
        if (fused) {
            if (s->var) { if (f.f64) residual_var_t<double, 0>(s, f, f.u, f.b, f.r); else residual_var_t<float, 0>(s, f, f.u, f.b, f.r); }
            else if (f.f64) launch_residual<double, 0>((const double*)f.u, (const double*)f.b, f.r, f.pitch, nullptr, nullptr, 1.0, f.N, f.pitch, 1, f.N, rpc, s->stream, -1, f.rows);
            else launch_residual<float, 0>((const float*)f.u, (const float*)f.b, f.r, f.pitch, nullptr, nullptr, 1.0, f.N, f.pitch, 1, f.N, rpc, s->stream, -1, f.rows);
            src = f.r;
        }
        if (mode >= MGX_RESTRICT_INJECT) {
            const double w = (mode == MGX_RESTRICT_INJECT4) ? 4.0 : 1.0;
            const dim3 blk(256), grd((c.N + 255) / 256, c.N - 1);
            if (f.f64) hipLaunchKernelGGL((k_restrict_inject<double>), grd, blk, 0, s->stream, (const double*)src, (double*)c.b,
                                          zero_guess ? (double*)c.u : nullptr, c.N, f.pitch, c.pitch, w);
            else hipLaunchKernelGGL((k_restrict_inject<float>), grd, blk, 0, s->stream, (const float*)src, (float*)c.b,
                                    zero_guess ? (float*)c.u : nullptr, c.N, f.pitch, c.pitch, (float)w);
        } else if (f.f64) {
            launch_restrict<double>((const double*)f.u, (const double*)src, (double*)c.b, zero_guess ? (double*)c.u : nullptr, f.N, f.pitch, c.pitch,
                                    1, c.N, 0, mode, false, rpc, s->stream);
        } else {
            launch_restrict<float>((const float*)f.u, (const float*)src, (float*)c.b, zero_guess ? (float*)c.u : nullptr, f.N, f.pitch, c.pitch,
                                   1, c.N, 0, mode, false, rpc, s->stream);
        }
        return;
    }
    if (f.f64)
        launch_restrict<double>((const double*)f.u, (const double*)f.b, (double*)c.b, zero_guess ? (double*)c.u : nullptr,
                                f.N, f.pitch, c.pitch, 1, c.N, 0, s->cfg.restrict_mode, fused, rpc, s->stream);
    else
        launch_restrict<float>((const float*)f.u, (const float*)f.b, (float*)c.b, zero_guess ? (float*)c.u : nullptr,
                               f.N, f.pitch, c.pitch, 1, c.N, 0, s->cfg.restrict_mode, fused, rpc, s->stream);
}

void prolong_level(mgx_solver* s, int level, bool add)
{
    Level& f = s->lv[level];
    Level& c = s->lv[level - 1];
    const bool fine = (level == s->cfg.finest_level);
    Prof p(s, fine ? MGX_PROF_PROLONG_FINE : MGX_PROF_COARSE, 1);
    const int rpc = s->rows_per_chunk;
    if (f.f64)
        launch_prolong<double>((double*)f.u, (const double*)c.u, f.N, f.pitch, c.pitch, 1, f.N, 0, add, rpc, s->stream);
    else
        launch_prolong<float>((float*)f.u, (const float*)c.u, f.N, f.pitch, c.pitch, 1, f.N, 0, add, rpc, s->stream);
}

void bottom_solve(mgx_solver* s)
{
    Level& l = s->lv[s->cfg.coarsest_level];
    Prof p(s, (l.L == s->cfg.finest_level) ? MGX_PROF_SMOOTH_FINE : MGX_PROF_COARSE, 4);
    if (s->var) {                                                        // MF:63-72: x = A^-1 b, A^-1 built at set-up
        const int n = l.N - 1;
        if (l.f64) hipLaunchKernelGGL((k_var_dense_solve<double>), dim3((n * n + 63) / 64), dim3(64), 0, s->stream, s->var_inv, (const double*)l.b, (double*)l.u, n, l.pitch);
        else hipLaunchKernelGGL((k_var_dense_solve<float>), dim3((n * n + 63) / 64), dim3(64), 0, s->stream, s->var_inv, (const float*)l.b, (float*)l.u, n, l.pitch);
        return;
    }
    if (l.f64) s->bottom.solve<double>((const double*)l.b, (double*)l.u, l.pitch, s->stream);
    else s->bottom.solve<float>((const float*)l.b, (float*)l.u, l.pitch, s->stream);
}

int zero_u(mgx_solver* s, int level);

// PS:575-627 / MF:132-173
void vcycle(mgx_solver* s, int level)
{
    if (level == s->cfg.coarsest_level) {
        if (s->cfg.bottom == MGX_BOTTOM_EXACT) {
            bottom_solve(s);                                  // MF:137-139
        } else {
            smooth(s, level, s->cfg.mu1);                     // PS:581
            smooth(s, level, s->cfg.mu2);                     // PS:585
        }
        return;
    }
    // Is this level's own iterate a known, not materialised, zero (set by the caller)?
    const bool zin_here = (s->zero_in_level == level);
    s->zero_in_level = -1;
    // PS:613: the coarse guess is zero.  If the coarse level's first pass can synthesise it,
    // nobody writes or reads those zeros.
    const bool zin_next = zero_in_ok(s, level - 1);
    if (zin_next) s->zero_in_level = level - 1;
    // PS:581 pre-smoothing + PS:604-613 residual, restriction, zero coarse guess:
    // one set of passes when the level is eligible for folding
    if (!smooth_folded(s, level, s->cfg.mu1, false, 1, zin_here)) {
        if (zin_here) (void)zero_u(s, level);                 // cannot happen (zero_in_ok == fold_eligible); stay correct
        smooth(s, level, s->cfg.mu1);                         // PS:581
        restrict_level(s, level, true, !zin_next);            // PS:604-613
    }
    const bool top_norm = s->want_norm && level == s->cfg.finest_level;
    s->want_norm = false;                                     // only the outermost level reports the norm
    vcycle(s, level - 1);                                     // PS:617
    // PS:620-624 correction + PS:625 post-smoothing (+ the cycle's residual norm)
    if (!smooth_folded(s, level, s->cfg.mu2, true, top_norm ? 2 : 0)) {
        prolong_level(s, level, true);                        // PS:620-624
        smooth(s, level, s->cfg.mu2);                         // PS:625
    }
}

int zero_u(mgx_solver* s, int level)
{
    Level& l = s->lv[level];
    HIPCHK(s, hipMemsetAsync(l.u, 0, l.bytes, s->stream));
    return MGX_OK;
}

// PS:629-650 / MF:175-191 on the working hierarchy (B[finest] must be set)
int fmg(mgx_solver* s)
{
    const int lo = s->cfg.coarsest_level, hi = s->cfg.finest_level;
    for (int l = hi; l > lo; --l) restrict_level(s, l, false, false);        // PS:641
    if (s->cfg.bottom == MGX_BOTTOM_EXACT) {
        bottom_solve(s);                                                     // MF:178-181
    } else {
        int rc = zero_u(s, lo);                                              // PS:630
        if (rc) return rc;
        for (int i = 0; i <= s->cfg.mu0; ++i) { s->zero_in_level = -1; vcycle(s, lo); }   // PS:635
    }
    for (int l = lo + 1; l <= hi; ++l) {
        prolong_level(s, l, false);                                          // PS:645
        for (int i = 0; i <= s->cfg.mu0; ++i) { s->zero_in_level = -1; vcycle(s, l); }    // PS:646-648
    }
    return MGX_OK;
}

// ||B - A U||^2 of an arbitrary grid pair (double or float) -> sum_host, enqueued only
int enqueue_norm(mgx_solver* s, const Level& l, const void* u, const void* b, int cls)
{
    if (s->norm_blocks_ready > 0 && u == s->lv[s->cfg.finest_level].u && !s->mixed) {
        // the last post-smoothing pass already summed (b - A u)^2 per block
        Prof p(s, cls, 1);
        hipLaunchKernelGGL(k_reduce_partials, dim3(1), dim3(kReduceThreads), 0, s->stream, s->partial, s->norm_blocks_ready,
                           s->sum_dev);
    } else if (s->var) {
        Prof p(s, cls, 2);
        if (l.f64) residual_var_t<double, 1>(s, l, u, b, nullptr); else residual_var_t<float, 1>(s, l, u, b, nullptr);
    } else {
        Prof p(s, cls, 2);
        if (l.f64)
            launch_residual<double, 1>((const double*)u, (const double*)b, nullptr, 0, s->partial, s->sum_dev, 1.0,
                                       l.N, l.pitch, 1, l.N, s->rows_per_chunk, s->stream, s->partial_cap, l.rows);
        else
            launch_residual<float, 1>((const float*)u, (const float*)b, nullptr, 0, s->partial, s->sum_dev, 1.0,
                                      l.N, l.pitch, 1, l.N, s->rows_per_chunk, s->stream, s->partial_cap, l.rows);
    }
    s->norm_blocks_ready = 0;
    HIPCHK(s, hipMemcpyAsync(s->sum_host, s->sum_dev, sizeof(double), hipMemcpyDeviceToHost, s->stream));
    return MGX_OK;
}

int residual_norm_grid(mgx_solver* s, const Level& l, const void* u, const void* b, double* out, int cls)
{
    int rc = enqueue_norm(s, l, u, b, cls);
    if (rc) return rc;
    HIPCHK(s, hipStreamSynchronize(s->stream));
    *out = std::sqrt(*s->sum_host);
    return MGX_OK;
}

// ---- mgx_solve's loop body: one V-cycle from the finest level + the residual norm -----------
// About 30 launches, the small levels launch-bound: with profiling off the whole body is
// captured once into a hipGraph and replayed.  The cycle swaps each level's u / tmp buffers
// on the host, so a graph is keyed by the buffer assignment it was captured with and carries
// the assignment it leaves behind (a cycle with an odd number of passes on some level
// alternates between two graphs).
std::vector<void*> buffer_state(const mgx_solver* s)
{
    std::vector<void*> v;
    for (int l = s->cfg.coarsest_level; l <= s->cfg.finest_level; ++l) { v.push_back(s->lv[l].u); v.push_back(s->lv[l].tmp); }
    return v;
}

void set_buffer_state(mgx_solver* s, const std::vector<void*>& v)
{
    size_t i = 0;
    for (int l = s->cfg.coarsest_level; l <= s->cfg.finest_level; ++l) { s->lv[l].u = v[i++]; s->lv[l].tmp = v[i++]; }
}

// want_norm: the body ends with the residual norm (mgx_solve's loop);  zero_start: the cycle
// starts from u = 0 (PS:613; the correction cycle of a coarse-grid solver), synthesised by the
// first pass where it can be, so nobody writes or reads the zeros
int cycle_body_direct(mgx_solver* s, bool want_norm, bool zero_start, double* r)
{
    const int L = s->cfg.finest_level;
    Level& l = s->lv[L];
    s->norm_blocks_ready = 0;
    s->want_norm = want_norm; s->zero_in_level = -1;
    if (zero_start && L > s->cfg.coarsest_level) {
        if (zero_in_ok(s, L)) s->zero_in_level = L;
        else { int rc = zero_u(s, L); if (rc) return rc; }
    }
    vcycle(s, L);
    s->want_norm = false;
    if (!want_norm) return MGX_OK;
    return residual_norm_grid(s, l, l.u, l.b, r, MGX_PROF_NORM_FINE);
}

// cfg.profile = 2: the finest level's passes are launched one by one between HIP events (they are
// long: the host runs ahead of them) and everything below the finest level is ONE graph replay
// between two events.  Events recorded inside a captured graph carry no time stamps on this runtime
// (hipEventElapsedTime: invalid resource handle - tools/probe/graph_events.hip), so this is how the
// dominant kernel is timed with HIP events while the cycle still runs the way mgx_solve runs it
// (cfg.profile = 1, every launch eager: 1.62 instead of 1.52 ms per cycle at 8192^2).
int coarse_part_graph(mgx_solver* s, int level)
{
    std::vector<void*> before = buffer_state(s);
    before.push_back(reinterpret_cast<void*>((uintptr_t)(4 | (s->zero_in_level == level ? 8 : 0))));
    mgx_solver::CycleGraph* g = nullptr;
    for (auto& c : s->graphs)
        if (c.before == before) { g = &c; break; }
    if (!g) {
        if (s->graphs.size() >= 8 || hipStreamBeginCapture(s->stream, hipStreamCaptureModeThreadLocal) != hipSuccess) {
            (void)hipGetLastError();
            vcycle(s, level);
            return MGX_OK;
        }
        const double fu0 = s->fine_updates;
        const std::vector<void*> state0 = buffer_state(s);
        const int zin0 = s->zero_in_level;
        s->prof_mute = true;
        vcycle(s, level);
        s->prof_mute = false;
        hipGraph_t graph = nullptr;
        hipGraphExec_t exec = nullptr;
        hipError_t e = hipStreamEndCapture(s->stream, &graph);
        if (e == hipSuccess) e = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
        if (graph) (void)hipGraphDestroy(graph);
        if (e != hipSuccess) {
            (void)hipGetLastError();
            set_buffer_state(s, state0);
            s->fine_updates = fu0;
            s->zero_in_level = zin0;
            s->use_graph = 0;
            vcycle(s, level);
            return MGX_OK;
        }
        mgx_solver::CycleGraph c;
        c.before = before; c.after = buffer_state(s); c.exec = exec; c.fine_updates = s->fine_updates - fu0;
        s->graphs.push_back(c);
        g = &s->graphs.back();
    } else {
        set_buffer_state(s, g->after);
        s->fine_updates += g->fine_updates;
    }
    s->zero_in_level = -1;
    Prof p(s, MGX_PROF_COARSE, 1);
    HIPCHK(s, hipGraphLaunch(g->exec, s->stream));
    return MGX_OK;
}

int cycle_body_split(mgx_solver* s, bool want_norm, bool zero_start, double* r)
{
    struct Chain {                      // spans share their boundary events while this cycle is enqueued (mgx_solver::prof_chain)
        mgx_solver* s;
        explicit Chain(mgx_solver* s_) : s(s_) { s->prof_chain = env_int("MGX_PROF_CHAIN", 1) != 0; s->chain_ev = nullptr; }
        ~Chain() { s->prof_chain = false; s->chain_ev = nullptr; }
    } chain(s);
    const int L = s->cfg.finest_level;
    Level& l = s->lv[L];
    s->norm_blocks_ready = 0;
    s->want_norm = false; s->zero_in_level = -1;
    bool zin_here = false;
    if (zero_start) {
        if (zero_in_ok(s, L)) zin_here = true;
        else { int rc = zero_u(s, L); if (rc) return rc; }
    }
    // the finest level's half of vcycle() (PS:581, 604-613), launch by launch
    const bool zin_next = zero_in_ok(s, L - 1);
    if (zin_next) s->zero_in_level = L - 1;                       // (the pre-smoothing pass then leaves the coarse guess alone)
    if (!smooth_folded(s, L, s->cfg.mu1, false, 1, zin_here)) {
        if (zin_here) (void)zero_u(s, L);
        smooth(s, L, s->cfg.mu1);
        restrict_level(s, L, true, !zin_next);
    }
    int rc = coarse_part_graph(s, L - 1);                         // PS:617
    if (rc) return rc;
    if (!smooth_folded(s, L, s->cfg.mu2, true, want_norm ? 2 : 0)) {   // PS:620-625 (+ the norm's sums)
        prolong_level(s, L, true);
        smooth(s, L, s->cfg.mu2);
    }
    if (!want_norm) return MGX_OK;
    return residual_norm_grid(s, l, l.u, l.b, r, MGX_PROF_NORM_FINE);
}

int cycle_body(mgx_solver* s, bool want_norm, bool zero_start, double* r)
{
    if (s->cfg.profile == 2 && s->use_graph && !s->mixed && s->cfg.finest_level > s->cfg.coarsest_level)
        return cycle_body_split(s, want_norm, zero_start, r);
    if (!s->use_graph || s->cfg.profile || s->mixed) return cycle_body_direct(s, want_norm, zero_start, r);
    const int L = s->cfg.finest_level;
    std::vector<void*> before = buffer_state(s);
    // the two flavours are different graphs: tag the key
    before.push_back(reinterpret_cast<void*>((uintptr_t)((want_norm ? 1 : 0) | (zero_start ? 2 : 0))));
    mgx_solver::CycleGraph* g = nullptr;
    for (auto& c : s->graphs)
        if (c.before == before) { g = &c; break; }
    if (!g) {
        if (s->graphs.size() >= 8) return cycle_body_direct(s, want_norm, zero_start, r);
        const double fu0 = s->fine_updates;
        const std::vector<void*> state0 = buffer_state(s);
        if (hipStreamBeginCapture(s->stream, hipStreamCaptureModeThreadLocal) != hipSuccess) {
            (void)hipGetLastError();
            s->use_graph = 0;
            return cycle_body_direct(s, want_norm, zero_start, r);
        }
        // enqueue the body (nothing executes while capturing; the host-side bookkeeping does)
        int rc = MGX_OK;
        s->norm_blocks_ready = 0;
        s->want_norm = want_norm; s->zero_in_level = -1;
        if (zero_start && L > s->cfg.coarsest_level) {
            if (zero_in_ok(s, L)) s->zero_in_level = L;
            else rc = zero_u(s, L);
        }
        vcycle(s, L);
        s->want_norm = false;
        if (rc == MGX_OK && want_norm) rc = enqueue_norm(s, s->lv[L], s->lv[L].u, s->lv[L].b, MGX_PROF_NORM_FINE);
        hipGraph_t graph = nullptr;
        hipGraphExec_t exec = nullptr;
        hipError_t e = hipStreamEndCapture(s->stream, &graph);
        if (rc == MGX_OK && e == hipSuccess) e = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
        if (graph) (void)hipGraphDestroy(graph);
        if (rc != MGX_OK || e != hipSuccess) {
            // undo the bookkeeping of the cycle that was not run, and stay on the direct path
            (void)hipGetLastError();
            set_buffer_state(s, state0);
            s->fine_updates = fu0;
            s->use_graph = 0;
            return cycle_body_direct(s, want_norm, zero_start, r);
        }
        mgx_solver::CycleGraph c;
        c.before = before; c.after = buffer_state(s); c.exec = exec; c.fine_updates = s->fine_updates - fu0;
        s->graphs.push_back(c);
        g = &s->graphs.back();
    } else {
        set_buffer_state(s, g->after);
        s->fine_updates += g->fine_updates;
    }
    s->norm_blocks_ready = 0;
    s->zero_in_level = -1;
    HIPCHK(s, hipGraphLaunch(g->exec, s->stream));
    if (!want_norm) return MGX_OK;
    HIPCHK(s, hipStreamSynchronize(s->stream));
    *r = std::sqrt(*s->sum_host);
    return MGX_OK;
}

inline int cycle_and_norm(mgx_solver* s, double* r) { return cycle_body(s, true, false, r); }

double pow2_floor(double x)
{
    int e;
    (void)std::frexp(x, &e);
    return std::ldexp(1.0, e - 1);
}

Level* pick(mgx_solver* s, int level, int which, void** grid)
{
    Level* l = &s->lv[level];
    if (s->mixed && level == s->cfg.finest_level) l = &s->fine64;
    switch (which) {
        case MGX_VEC_U: *grid = l->u; break;
        case MGX_VEC_B: *grid = l->b; break;
        case MGX_VEC_R: *grid = l->r; break;
        default: *grid = nullptr;
    }
    return l;
}

// non-homogeneous Dirichlet data: b_ij += g of the boundary neighbours (mgx_set_rhs_dirichlet)
template <typename T>
void fold_ring(std::vector<T>& b, const T* ring, size_t n)
{
    const size_t N = n + 1;
    const T* top = ring;                 // row 0, columns 0..N
    const T* bot = ring + (N + 1);       // row N, columns 0..N
    const T* lef = ring + 2 * (N + 1);   // column 0, rows 1..N-1
    const T* rig = lef + (N - 1);        // column N, rows 1..N-1
    for (size_t j = 0; j < n; ++j) {
        b[j] += top[j + 1];                          // interior row 1 touches boundary row 0
        b[(n - 1) * n + j] += bot[j + 1];            // interior row N-1 touches boundary row N
    }
    for (size_t i = 0; i < n; ++i) {
        b[i * n] += lef[i];                          // interior column 1 touches boundary column 0
        b[i * n + (n - 1)] += rig[i];
    }
}
} // namespace

#include "mgx_dist.hpp"

// entry points that make no sense on a multi-GPU handle
#define NO_DIST(s)                                                                        \
    if ((s)->dist) return (s)->fail(MGX_ERR_STATE, "not available on a multi-GPU handle (see mgx.h, Multi-GPU)");

// =====================================================================================
// C-ABI
// =====================================================================================
extern "C" {

int mgx_config_default(mgx_config* c)
{
    if (!c) return MGX_ERR_INVALID;
    c->finest_level = 10;      // PS:17
    c->coarsest_level = 7;     // PS:18
    c->mu0 = 30;               // PS:20
    c->mu1 = 10;               // PS:21
    c->mu2 = 10;               // PS:22
    c->omega = 2.0 / 3.0;      // PS:127
    c->smoother = MGX_SMOOTHER_JACOBI;
    c->dtype = MGX_DTYPE_F64;
    c->schedule = MGX_SCHEDULE_FMG;   // PS:727
    c->restrict_mode = MGX_RESTRICT_CONSISTENT;
    c->bottom = MGX_BOTTOM_EXACT;
    c->device = 0;
    c->profile = 0;
    c->n_gpus = 0;             // PS:659: one queue
    c->cut_level = 0;
    for (int i = 0; i < MGX_MAX_GPUS; ++i) c->devices[i] = -1;
    c->arith = MGX_ARITH_SEPARATE;
    c->op = MGX_OPERATOR_POISSON;
    return MGX_OK;
}

const char* mgx_status_string(int st)
{
    switch (st) {
        case MGX_OK: return "ok";
        case MGX_ERR_INVALID: return "invalid argument";
        case MGX_ERR_NO_DEVICE: return "no usable HIP device";
        case MGX_ERR_HIP: return "HIP runtime error";
        case MGX_ERR_ALLOC: return "device allocation failed";
        case MGX_ERR_STATE: return "invalid state";
        default: return "unknown status";
    }
}

const char* mgx_last_error(mgx_handle h) { return h ? h->err.c_str() : g_create_error.c_str(); }

int mgx_level_n(int level) { return (level >= 1 && level < 31) ? (1 << level) - 1 : -1; }

long mgx_level_pitch(int level, int dtype)
{
    if (level < 1 || level > 20 || (dtype != MGX_DTYPE_F32 && dtype != MGX_DTYPE_F64)) return -1;
    return level_pitch(level, dtype);
}

int mgx_create(const mgx_config* cfg, mgx_handle* out)
{
    if (!cfg || !out) { g_create_error = "null argument"; return MGX_ERR_INVALID; }
    *out = nullptr;
    // level 2 (N = 4) is the smallest grid whose rows hold whole 16-byte vectors
    // in both precisions; level 8 (255^2) bounds the dense sine-transform solve.
    if (cfg->coarsest_level < 2 || cfg->finest_level < cfg->coarsest_level || cfg->finest_level > 15 ||
        cfg->mu0 < 0 || cfg->mu1 < 0 || cfg->mu2 < 0 || !(cfg->omega > 0.0 && cfg->omega < 2.0) ||
        cfg->smoother < 0 || cfg->smoother > 1 || cfg->dtype < 0 || cfg->dtype > 2 ||
        cfg->schedule < 0 || cfg->schedule > 1 || cfg->restrict_mode < 0 || cfg->restrict_mode > MGX_RESTRICT_INJECT4 ||
        cfg->bottom < 0 || cfg->bottom > 1 || cfg->arith < 0 || cfg->arith > 1 || cfg->op < 0 || cfg->op > 1) {
        g_create_error = "invalid configuration";
        return MGX_ERR_INVALID;
    }
    const bool var = (cfg->op == MGX_OPERATOR_STENCIL5);
    if (cfg->bottom == MGX_BOTTOM_EXACT && cfg->coarsest_level > (var ? 5 : 8)) {
        g_create_error = var ? "exact bottom solve of a general operator (dense inverse) supports coarsest_level <= 5"
                             : "exact bottom solve supports coarsest_level <= 8";
        return MGX_ERR_INVALID;
    }
    if (var && (cfg->dtype == MGX_DTYPE_MIXED || cfg->smoother != MGX_SMOOTHER_JACOBI || cfg->arith != MGX_ARITH_SEPARATE || cfg->n_gpus > 1)) {
        g_create_error = "MGX_OPERATOR_STENCIL5: dtype F64 or F32, Jacobi (MF:75-96), arith SEPARATE, one GPU";
        return MGX_ERR_INVALID;
    }
    if (cfg->restrict_mode >= MGX_RESTRICT_INJECT && (cfg->n_gpus > 1 || cfg->dtype == MGX_DTYPE_MIXED)) {
        g_create_error = "injection restriction (MF:122-130): single-GPU handles of dtype F64 or F32";
        return MGX_ERR_INVALID;
    }
    if (cfg->n_gpus > 1) return mgx_create_rank(cfg, -1, 1, nullptr, nullptr, out);
    log_runtime_libs("mgx_create");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || cfg->device < 0 || cfg->device >= ndev) {
        g_create_error = "no usable HIP device (libmgx has no CPU fallback)";
        return MGX_ERR_NO_DEVICE;
    }
    if (hipSetDevice(cfg->device) != hipSuccess) { g_create_error = "hipSetDevice failed"; return MGX_ERR_HIP; }

    mgx_solver* s = new (std::nothrow) mgx_solver();
    if (!s) { g_create_error = "out of host memory"; return MGX_ERR_ALLOC; }
    s->cfg = *cfg;
    s->mixed = (cfg->dtype == MGX_DTYPE_MIXED);
    s->var = var;
    s->work_f64 = (cfg->dtype == MGX_DTYPE_F64);
    s->rows_per_chunk = env_int("MGX_ROWS", 0);
    s->fuse = fuse_cfg();
    s->fuse.arith = cfg->arith;
    s->fold = env_int("MGX_FOLD", 1);
    s->use_zero_in = env_int("MGX_ZERO_IN", 1);
    s->use_graph = env_int("MGX_GRAPH", 1);
    int rc = MGX_OK;
    auto bail = [&](int code) { g_create_error = s->err; mgx_destroy(s); return code; };
    if (hipStreamCreate(&s->stream) != hipSuccess) { s->err = "hipStreamCreate failed"; return bail(MGX_ERR_HIP); }
    s->lv.resize(cfg->finest_level + 1);
    for (int l = cfg->coarsest_level; l <= cfg->finest_level; ++l)
        if ((rc = alloc_level(s, s->lv[l], l, s->work_f64)) != MGX_OK) return bail(rc);
    if (s->mixed && (rc = alloc_level(s, s->fine64, cfg->finest_level, true)) != MGX_OK) return bail(rc);
    if (var || cfg->restrict_mode >= MGX_RESTRICT_INJECT) {
        // these cycles form the residual as a grid (MF:150-153) before restricting it: allocate it now (nothing may be
        // allocated while a cycle is being captured into a graph)
        for (int l = cfg->coarsest_level + 1; l <= cfg->finest_level; ++l)
            if ((rc = ensure_r(s, s->lv[l])) != MGX_OK) return bail(rc);
    }
    if (var) {
        for (int l = cfg->coarsest_level; l <= cfg->finest_level; ++l)
            if ((rc = var_alloc_level(s, s->lv[l])) != MGX_OK) return bail(rc);
        if (cfg->bottom == MGX_BOTTOM_EXACT) {
            const size_t NN = (size_t)((1 << cfg->coarsest_level) - 1) * ((1 << cfg->coarsest_level) - 1);
            if (hipMalloc(&s->var_M, NN * NN * sizeof(double)) != hipSuccess || hipMalloc(&s->var_inv, NN * NN * sizeof(double)) != hipSuccess ||
                hipMalloc(&s->var_pm, NN * sizeof(double)) != hipSuccess || hipMalloc(&s->var_pi, NN * sizeof(double)) != hipSuccess) {
                s->err = "allocation of the coarsest operator's dense inverse failed";
                return bail(MGX_ERR_ALLOC);
            }
        }
    }
    s->mixed_fuse = env_int("MGX_MIXED_FUSE", 1);
    // fine64.tmp: the out-of-place target of the fused update + residual pass
    if (s->mixed && !s->mixed_fuse) { (void)hipFree(s->fine64.tmp); s->fine64.tmp = nullptr; }
    // partial sums: the largest launch any norm kernel can make on the finest level
    {
        const int N = 1 << cfg->finest_level;
        long cap = 0;
        for (int rpc : {s->rows_per_chunk, 1}) {
            cap = std::max(cap, sumsq_blocks<double>(N, N, rpc));
            cap = std::max(cap, sumsq_blocks<float>(N, N, rpc));
        }
        if (N <= s->fuse.tile_max_n) cap = std::max(cap, (long)((N + 31) / 32) * ((N + 39) / 40));   // tiles >= 32 x 40
        s->partial_cap = cap + 8;
        if (hipMalloc(&s->partial, s->partial_cap * sizeof(double)) != hipSuccess ||
            hipMalloc(&s->sum_dev, sizeof(double)) != hipSuccess ||
            hipHostMalloc(&s->sum_host, sizeof(double)) != hipSuccess) {
            s->err = "allocation of reduction buffers failed";
            return bail(MGX_ERR_ALLOC);
        }
    }
    if (cfg->bottom == MGX_BOTTOM_EXACT && !var) {
        if (s->bottom.init((1 << cfg->coarsest_level) - 1) != hipSuccess) {
            s->err = "bottom solver allocation failed";
            return bail(MGX_ERR_ALLOC);
        }
    }
    if (hipStreamSynchronize(s->stream) != hipSuccess) { s->err = "device initialisation failed"; return bail(MGX_ERR_HIP); }
    *out = s;
    return MGX_OK;
}

int mgx_destroy(mgx_handle s)
{
    if (!s) return MGX_OK;
    if (s->dist) { dist_free(s->dist); s->dist = nullptr; }
    if (s->stream) (void)hipStreamSynchronize(s->stream);
    for (auto& l : s->lv) free_level(l);
    free_level(s->fine64);
    s->bottom.destroy();
    if (s->partial) (void)hipFree(s->partial);
    if (s->sum_dev) (void)hipFree(s->sum_dev);
    if (s->sum_host) (void)hipHostFree(s->sum_host);
    for (double* p : {s->var_M, s->var_inv, s->var_pm, s->var_pi}) if (p) (void)hipFree(p);
    for (auto& g : s->graphs) if (g.exec) (void)hipGraphExecDestroy(g.exec);
    for (auto& p : s->ev_used) { (void)hipEventDestroy(p.a); (void)hipEventDestroy(p.b); }
    for (auto& p : s->ev_free) { (void)hipEventDestroy(p.a); (void)hipEventDestroy(p.b); }
    if (s->stream) (void)hipStreamDestroy(s->stream);
    delete s;
    return MGX_OK;
}

int mgx_graphs_cached(mgx_handle s)
{
    if (!s) return MGX_ERR_INVALID;
    if (s->dist) {
        // the replicated coarse levels of the first local slab replay their cycle from a graph
        mgx_handle c = s->dist->slabs[0].coarse;
        return (c && c->use_graph) ? (int)c->graphs.size() : -1;
    }
    if (!s->use_graph || s->cfg.profile == 1 || s->mixed) return -1;
    return (int)s->graphs.size();
}

int mgx_synchronize(mgx_handle s)
{
    if (!s) return MGX_ERR_INVALID;
    if (s->dist) return dist_sync(s, s->dist);
    HIPCHK(s, hipStreamSynchronize(s->stream));
    return MGX_OK;
}

// ---- data ---------------------------------------------------------------------------
int mgx_set_level(mgx_handle s, int level, int which, const void* src, size_t count)
{
    if (!s || !src) return MGX_ERR_INVALID;
    if (!level_ok(s, level)) return s->fail(MGX_ERR_INVALID, "level out of range");
    if (s->dist) {
        if (level != s->cfg.finest_level || (which != MGX_VEC_U && which != MGX_VEC_B))
            return s->fail(MGX_ERR_STATE, "multi-GPU handles exchange U and B of the finest level only");
        return dist_set(s, s->dist, which, src, count);
    }
    void* grid = nullptr;
    Level* l = pick(s, level, which, &grid);
    if (which == MGX_VEC_R) {
        int rc = ensure_r(s, *l);
        if (rc) return rc;
        grid = l->r;
    }
    if (!grid) return s->fail(MGX_ERR_INVALID, "unknown vector selector");
    return copy_in(s, *l, grid, src, count);
}

int mgx_get_level(mgx_handle s, int level, int which, void* dst, size_t count)
{
    if (!s || !dst) return MGX_ERR_INVALID;
    if (!level_ok(s, level)) return s->fail(MGX_ERR_INVALID, "level out of range");
    if (s->dist) {
        if (level != s->cfg.finest_level || (which != MGX_VEC_U && which != MGX_VEC_B))
            return s->fail(MGX_ERR_STATE, "multi-GPU handles exchange U and B of the finest level only");
        return dist_get(s, s->dist, which, dst, count);
    }
    void* grid = nullptr;
    Level* l = pick(s, level, which, &grid);
    if (!grid) return s->fail(MGX_ERR_STATE, "vector not available (call mgx_residual first for MGX_VEC_R)");
    return copy_out(s, *l, grid, dst, count);
}

int mgx_set_level_device(mgx_handle s, int level, int which, const void* grid)
{
    if (!s || !grid) return MGX_ERR_INVALID;
    NO_DIST(s)
    if (!level_ok(s, level)) return s->fail(MGX_ERR_INVALID, "level out of range");
    void* dst = nullptr;
    Level* l = pick(s, level, which, &dst);
    if (which == MGX_VEC_R) {
        int rc = ensure_r(s, *l);
        if (rc) return rc;
        dst = l->r;
    }
    if (!dst) return s->fail(MGX_ERR_INVALID, "unknown vector selector");
    HIPCHK(s, hipMemcpyAsync(dst, grid, l->bytes, hipMemcpyDeviceToDevice, s->stream));
    HIPCHK(s, hipStreamSynchronize(s->stream));
    return MGX_OK;
}

int mgx_get_level_device(mgx_handle s, int level, int which, void* grid)
{
    if (!s || !grid) return MGX_ERR_INVALID;
    NO_DIST(s)
    if (!level_ok(s, level)) return s->fail(MGX_ERR_INVALID, "level out of range");
    void* src = nullptr;
    Level* l = pick(s, level, which, &src);
    if (!src) return s->fail(MGX_ERR_STATE, "vector not available");
    HIPCHK(s, hipMemcpyAsync(grid, src, l->bytes, hipMemcpyDeviceToDevice, s->stream));
    HIPCHK(s, hipStreamSynchronize(s->stream));
    return MGX_OK;
}

int mgx_zero_level(mgx_handle s, int level, int which)
{
    if (!s) return MGX_ERR_INVALID;
    if (!level_ok(s, level)) return s->fail(MGX_ERR_INVALID, "level out of range");
    if (s->dist) {
        if (level != s->cfg.finest_level || which != MGX_VEC_U) return s->fail(MGX_ERR_STATE, "multi-GPU handles: only U of the finest level");
        return dist_zero_u(s, s->dist);
    }
    void* dst = nullptr;
    Level* l = pick(s, level, which, &dst);
    if (!dst) return s->fail(MGX_ERR_STATE, "vector not available");
    HIPCHK(s, hipMemsetAsync(dst, 0, l->bytes, s->stream));
    HIPCHK(s, hipStreamSynchronize(s->stream));
    return MGX_OK;
}

int mgx_set_rhs(mgx_handle s, const void* b, size_t count) { return s ? mgx_set_level(s, s->cfg.finest_level, MGX_VEC_B, b, count) : MGX_ERR_INVALID; }
int mgx_set_rhs_dirichlet(mgx_handle s, const void* b, size_t count, const void* ring, size_t ring_count)
{
    if (!s || !b || !ring) return MGX_ERR_INVALID;
    const int L = s->cfg.finest_level;
    const size_t n = (size_t)((1 << L) - 1);
    if (count != n * n) return s->fail(MGX_ERR_INVALID, "vector length must be n*n with n = 2^level - 1");
    if (ring_count != 4 * (n + 1)) return s->fail(MGX_ERR_INVALID, "ring must hold 4 N boundary values, N = n + 1");
    const bool f64 = (s->cfg.dtype != MGX_DTYPE_F32);
    if (f64) {
        std::vector<double> t((const double*)b, (const double*)b + count);
        fold_ring<double>(t, (const double*)ring, n);
        return mgx_set_level(s, L, MGX_VEC_B, t.data(), count);
    }
    std::vector<float> t((const float*)b, (const float*)b + count);
    fold_ring<float>(t, (const float*)ring, n);
    return mgx_set_level(s, L, MGX_VEC_B, t.data(), count);
}

int mgx_set_guess(mgx_handle s, const void* u, size_t count) { return s ? mgx_set_level(s, s->cfg.finest_level, MGX_VEC_U, u, count) : MGX_ERR_INVALID; }
int mgx_get_solution(mgx_handle s, void* u, size_t count) { return s ? mgx_get_level(s, s->cfg.finest_level, MGX_VEC_U, u, count) : MGX_ERR_INVALID; }

int mgx_fill_rhs(mgx_handle s, int kind, double f)
{
    if (!s) return MGX_ERR_INVALID;
    if (kind < 0 || kind > 1) return s->fail(MGX_ERR_INVALID, "unknown rhs kind");
    if (s->dist) return dist_fill(s, s->dist, MGX_VEC_B, kind, f, 0);
    void* grid = nullptr;
    Level* l = pick(s, s->cfg.finest_level, MGX_VEC_B, &grid);
    const dim3 blk(256), grd((l->N + 1 + 255) / 256, l->N + 1);
    if (l->f64) hipLaunchKernelGGL(k_fill_rhs<double>, grd, blk, 0, s->stream, (double*)grid, l->N, l->pitch, kind, f);
    else hipLaunchKernelGGL(k_fill_rhs<float>, grd, blk, 0, s->stream, (float*)grid, l->N, l->pitch, kind, f);
    HIPCHK(s, hipStreamSynchronize(s->stream));
    return MGX_OK;
}

int mgx_fill_guess_random(mgx_handle s, uint64_t seed)
{
    if (!s) return MGX_ERR_INVALID;
    if (s->dist) return dist_fill(s, s->dist, MGX_VEC_U, 0, 0.0, seed);
    void* grid = nullptr;
    Level* l = pick(s, s->cfg.finest_level, MGX_VEC_U, &grid);
    const dim3 blk(256), grd((l->N + 1 + 255) / 256, l->N + 1);
    if (l->f64) hipLaunchKernelGGL(k_fill_random<double>, grd, blk, 0, s->stream, (double*)grid, l->N, l->pitch, seed);
    else hipLaunchKernelGGL(k_fill_random<float>, grd, blk, 0, s->stream, (float*)grid, l->N, l->pitch, seed);
    HIPCHK(s, hipStreamSynchronize(s->stream));
    return MGX_OK;
}

// ---- general per-level operators (MF:16-41) ---------------------------------------------------
int mgx_set_stencil(mgx_handle s, int level, const void* c, const void* n, const void* so, const void* w, const void* e, size_t count)
{
    if (!s || !c || !n || !so || !w || !e) return MGX_ERR_INVALID;
    NO_DIST(s)
    if (!s->var) return s->fail(MGX_ERR_STATE, "handle was created with op = MGX_OPERATOR_POISSON (the constant stencil needs no coefficients)");
    if (!level_ok(s, level)) return s->fail(MGX_ERR_INVALID, "level out of range");
    Level& l = s->lv[level];
    const void* src[5] = {c, n, so, w, e};
    for (int q = 0; q < 5; ++q) {
        int rc = copy_in(s, l, l.coef[q], src[q], count);
        if (rc) return rc;
    }
    return var_build(s, l);
}

int mgx_set_coefficient(mgx_handle s, const double* a_nodes, size_t count)
{
    if (!s || !a_nodes) return MGX_ERR_INVALID;
    NO_DIST(s)
    if (!s->var) return s->fail(MGX_ERR_STATE, "handle was created with op = MGX_OPERATOR_POISSON");
    const int Lf = s->cfg.finest_level, Nf = 1 << Lf;
    if (count != (size_t)(Nf + 1) * (size_t)(Nf + 1)) return s->fail(MGX_ERR_INVALID, "coefficient must hold (N + 1)^2 nodal values, N = 2^finest_level");
    double* dev = nullptr;
    if (hipMalloc(&dev, count * sizeof(double)) != hipSuccess) return s->fail(MGX_ERR_ALLOC, "hipMalloc failed for the nodal coefficient");
    int rc = MGX_OK;
    if (hipMemcpyAsync(dev, a_nodes, count * sizeof(double), hipMemcpyHostToDevice, s->stream) != hipSuccess) rc = s->fail(MGX_ERR_HIP, "copy of the nodal coefficient failed");
    for (int lv = s->cfg.coarsest_level; lv <= Lf && rc == MGX_OK; ++lv) {
        Level& l = s->lv[lv];
        const dim3 blk(256), grd((l.N + 1 + 255) / 256, l.N + 1);
        const int q = 1 << (Lf - lv);
        if (l.f64) hipLaunchKernelGGL((k_var_from_nodes<double>), grd, blk, 0, s->stream, dev, Nf, q, (double*)l.coef[0], (double*)l.coef[1],
                                      (double*)l.coef[2], (double*)l.coef[3], (double*)l.coef[4], l.N, l.pitch);
        else hipLaunchKernelGGL((k_var_from_nodes<float>), grd, blk, 0, s->stream, dev, Nf, q, (float*)l.coef[0], (float*)l.coef[1],
                                (float*)l.coef[2], (float*)l.coef[3], (float*)l.coef[4], l.N, l.pitch);
        rc = var_build(s, l);
    }
    (void)hipStreamSynchronize(s->stream);
    (void)hipFree(dev);
    return rc;
}

int mgx_get_stencil(mgx_handle s, int level, int which, void* dst, size_t count)
{
    if (!s || !dst) return MGX_ERR_INVALID;
    NO_DIST(s)
    if (!s->var) return s->fail(MGX_ERR_STATE, "handle was created with op = MGX_OPERATOR_POISSON");
    if (!level_ok(s, level) || which < 0 || which > 9) return s->fail(MGX_ERR_INVALID, "level or array selector out of range");
    Level& l = s->lv[level];
    if (!l.stencil_set) return s->fail(MGX_ERR_STATE, "operator of this level not set");
    return copy_out(s, l, which < 5 ? l.coef[which] : l.jac[which - 5], dst, count);
}

// ---- operators ------------------------------------------------------------------------
// On a MIXED handle the finest level exists twice: the double u, b the accessors address
// (mgx_set_level / mgx_get_level / mgx_set_rhs ...) and the float correction / residual scratch of
// the inner cycle, which is what the working hierarchy holds at that level.  An operator or
// schedule call there would silently act on the scratch pair: refuse it (only mgx_solve, and
// operators on the coarser float levels, are meaningful on a MIXED handle).
#define MIXED_GUARD(lvl)                                                                  \
    if (s->mixed && (lvl) == s->cfg.finest_level)                                         \
        return s->fail(MGX_ERR_STATE, "dtype MIXED: operators and schedules are not defined on the finest level " \
                                      "(double data, float inner cycle); use mgx_solve, or a F64 / F32 handle");

#define OP_PROLOGUE(lvl_min)                                                              \
    if (!s) return MGX_ERR_INVALID;                                                       \
    NO_DIST(s)                                                                            \
    if (level < (lvl_min) || level > s->cfg.finest_level)                                 \
        return s->fail(MGX_ERR_INVALID, "level out of range for this operator");          \
    MIXED_GUARD(level)                                                                    \
    if (int vr__ = var_ready(s, s->cfg.coarsest_level, level)) return vr__;

#define OP_EPILOGUE                                                                       \
    HIPCHK(s, hipGetLastError());                                                         \
    HIPCHK(s, hipStreamSynchronize(s->stream));                                           \
    return MGX_OK;

int mgx_smooth(mgx_handle s, int level, int mu)
{
    OP_PROLOGUE(s->cfg.coarsest_level)
    if (mu < 0) return s->fail(MGX_ERR_INVALID, "mu must be >= 0");
    smooth(s, level, mu);
    OP_EPILOGUE
}

int mgx_residual(mgx_handle s, int level)
{
    OP_PROLOGUE(s->cfg.coarsest_level)
    Level& l = s->lv[level];
    int rc = ensure_r(s, l);
    if (rc) return rc;
    if (s->var) {
        if (l.f64) residual_var_t<double, 0>(s, l, l.u, l.b, l.r); else residual_var_t<float, 0>(s, l, l.u, l.b, l.r);
    } else if (l.f64)
        launch_residual<double, 0>((const double*)l.u, (const double*)l.b, l.r, l.pitch, nullptr, nullptr, 1.0, l.N,
                                   l.pitch, 1, l.N, s->rows_per_chunk, s->stream, -1, l.rows);
    else
        launch_residual<float, 0>((const float*)l.u, (const float*)l.b, l.r, l.pitch, nullptr, nullptr, 1.0, l.N,
                                  l.pitch, 1, l.N, s->rows_per_chunk, s->stream, -1, l.rows);
    OP_EPILOGUE
}

int mgx_restrict(mgx_handle s, int level)
{
    OP_PROLOGUE(s->cfg.coarsest_level + 1)
    restrict_level(s, level, true, true);
    OP_EPILOGUE
}

int mgx_restrict_rhs(mgx_handle s, int level)
{
    OP_PROLOGUE(s->cfg.coarsest_level + 1)
    restrict_level(s, level, false, false);
    OP_EPILOGUE
}

int mgx_prolong_add(mgx_handle s, int level)
{
    OP_PROLOGUE(s->cfg.coarsest_level + 1)
    prolong_level(s, level, true);
    OP_EPILOGUE
}

int mgx_prolong(mgx_handle s, int level)
{
    OP_PROLOGUE(s->cfg.coarsest_level + 1)
    prolong_level(s, level, false);
    OP_EPILOGUE
}

int mgx_bottom_solve(mgx_handle s)
{
    if (!s) return MGX_ERR_INVALID;
    NO_DIST(s)
    if (s->cfg.bottom != MGX_BOTTOM_EXACT) return s->fail(MGX_ERR_STATE, "handle was created with bottom = SMOOTH");
    MIXED_GUARD(s->cfg.coarsest_level)
    if (int vr = var_ready(s, s->cfg.coarsest_level, s->cfg.coarsest_level)) return vr;
    bottom_solve(s);
    OP_EPILOGUE
}

int mgx_residual_norm(mgx_handle s, int level, double* out)
{
    if (!s || !out) return MGX_ERR_INVALID;
    if (!level_ok(s, level)) return s->fail(MGX_ERR_INVALID, "level out of range");
    if (s->dist) {
        if (level != s->cfg.finest_level) return s->fail(MGX_ERR_STATE, "multi-GPU handles: residual norm of the finest level only");
        return dist_norm(s, s->dist, out);
    }
    if (int vr = var_ready(s, level, level)) return vr;
    void* gu = nullptr; void* gb = nullptr;
    Level* l = pick(s, level, MGX_VEC_U, &gu);
    (void)pick(s, level, MGX_VEC_B, &gb);
    return residual_norm_grid(s, *l, gu, gb, out, level == s->cfg.finest_level ? MGX_PROF_NORM_FINE : MGX_PROF_COARSE);
}

int mgx_vcycle(mgx_handle s, int level)
{
    if (s && s->dist) {
        if (level != s->cfg.finest_level) return s->fail(MGX_ERR_STATE, "multi-GPU handles: V-cycles from the finest level only");
        int rc = dist_vcycle(s, s->dist);
        return rc ? rc : dist_sync(s, s->dist);
    }
    OP_PROLOGUE(s->cfg.coarsest_level)
    s->zero_in_level = -1;
    vcycle(s, level);
    OP_EPILOGUE
}

int mgx_vcycle_zero(mgx_handle s)
{
    if (!s) return MGX_ERR_INVALID;
    NO_DIST(s)
    MIXED_GUARD(s->cfg.finest_level)
    if (int vr = var_ready(s, s->cfg.coarsest_level, s->cfg.finest_level)) return vr;
    double unused = 0.0;
    int rc = cycle_body(s, false, true, &unused);
    if (rc) return rc;
    OP_EPILOGUE
}

int mgx_fmg(mgx_handle s)
{
    if (!s) return MGX_ERR_INVALID;
    if (s->dist) {
        int rc = dist_fmg(s, s->dist);
        return rc ? rc : dist_sync(s, s->dist);
    }
    MIXED_GUARD(s->cfg.finest_level)
    if (int vr = var_ready(s, s->cfg.coarsest_level, s->cfg.finest_level)) return vr;
    int rc = fmg(s);
    if (rc) return rc;
    OP_EPILOGUE
}

// ---- solve ----------------------------------------------------------------------------
int mgx_solve(mgx_handle s, double tol, int max_cycles, mgx_stats* stats, double* history, int history_cap)
{
    if (!s || max_cycles < 0 || !(tol >= 0.0)) return MGX_ERR_INVALID;
    if (s->dist) return dist_solve(s, s->dist, tol, max_cycles, stats, history, history_cap);
    if (int vr = var_ready(s, s->cfg.coarsest_level, s->cfg.finest_level)) return vr;
    const int L = s->cfg.finest_level;
    const bool do_fmg = (s->cfg.schedule == MGX_SCHEDULE_FMG);
    std::vector<double> hist;
    hist.reserve(max_cycles + 1);
    s->fine_updates = 0.0;
    HIPCHK(s, hipStreamSynchronize(s->stream));
    const auto t0 = std::chrono::steady_clock::now();
    int rc = MGX_OK;
    double r = 0.0;
    int k = 0;

    if (!s->mixed) {
        Level& l = s->lv[L];
        if ((rc = residual_norm_grid(s, l, l.u, l.b, &r, MGX_PROF_NORM_FINE))) return rc;
        hist.push_back(r);
        for (k = 0; k < max_cycles; ++k) {
            if (hist[k] <= tol * hist[0]) break;
            s->norm_blocks_ready = 0;
            if (k == 0 && do_fmg) {
                if ((rc = fmg(s))) return rc;
                if ((rc = residual_norm_grid(s, l, l.u, l.b, &r, MGX_PROF_NORM_FINE))) return rc;
            } else if ((rc = cycle_and_norm(s, &r))) {
                return rc;
            }
            hist.push_back(r);
        }
    } else {
        // BASELINE config 5: double residual and solution, float inner cycle on
        // the residual scaled by a power of two near its rms (exact scaling).
        Level& w = s->lv[L];            // float: u = e32, b = r32
        Level& d = s->fine64;           // double: u, b
        const double n = (double)(w.N - 1);
        const int rpc = s->rows_per_chunk;
        if ((rc = residual_norm_grid(s, d, d.u, d.b, &r, MGX_PROF_NORM_FINE))) return rc;
        hist.push_back(r);
        for (k = 0; k < max_cycles; ++k) {
            if (hist[k] <= tol * hist[0]) break;
            const Launch g = make_launch(d.N, 2, d.N - 1, rpc);
            double pending_scale = 0.0;
            if (k == 0 && do_fmg) {
                // ||b||: residual norm against a zero guess (FMG discards the guess, PS:630)
                HIPCHK(s, hipMemsetAsync(d.u, 0, d.bytes, s->stream));
                double bn = 0.0;
                if ((rc = residual_norm_grid(s, d, d.u, d.b, &bn, MGX_PROF_NORM_FINE))) return rc;
                const double scale = pow2_floor(bn / n);
                hipLaunchKernelGGL(k_scale_f64_to_f32, dim3(g.blocks), dim3(kBlock), 0, s->stream, (float*)w.b,
                                   (const double*)d.b, 1.0 / scale, d.N, d.pitch, w.pitch, 1, d.N, g.R, g.strips, g.chunks);
                if ((rc = fmg(s))) return rc;
                hipLaunchKernelGGL(k_axpy_f32_to_f64, dim3(g.blocks), dim3(kBlock), 0, s->stream, (double*)d.u,
                                   (const float*)w.u, scale, d.N, d.pitch, w.pitch, 1, d.N, g.R, g.strips, g.chunks, 1);
            } else {
                const double scale = pow2_floor(hist[k > 0 ? k - 1 : 0] / n);
                if (k == 0) {
                    // no scaled residual is pending yet: produce it now
                    Prof p(s, MGX_PROF_NORM_FINE, 2);
                    launch_residual<double, 2>((const double*)d.u, (const double*)d.b, w.b, w.pitch, s->partial,
                                               s->sum_dev, 1.0 / scale, d.N, d.pitch, 1, d.N, rpc, s->stream, s->partial_cap, d.rows);
                }
                // PS:613-style zero guess: implicit when the first pass can synthesise it
                if (zero_in_ok(s, L)) s->zero_in_level = L;
                else HIPCHK(s, hipMemsetAsync(w.u, 0, w.bytes, s->stream));
                vcycle(s, L);
                pending_scale = scale;                        // u += scale * e still to be applied
                if (!(s->mixed_fuse && d.tmp)) {
                    hipLaunchKernelGGL(k_axpy_f32_to_f64, dim3(g.blocks), dim3(kBlock), 0, s->stream, (double*)d.u,
                                       (const float*)w.u, scale, d.N, d.pitch, w.pitch, 1, d.N, g.R, g.strips, g.chunks, 0);
                    pending_scale = 0.0;
                }
            }
            // residual of the new iterate: its norm is hist[k+1]; the same pass
            // writes the float residual the next cycle consumes, scaled by the
            // power of two derived from hist[k] (known now).
            {
                const double next_scale = pow2_floor(hist[k] / n);
                Prof p(s, MGX_PROF_NORM_FINE, 2);
                if (pending_scale != 0.0) {
                    // the correction and the residual in one out-of-place pass (32 instead of 40 B per point)
                    // rows per wave of this pass (MGX_MIXED_ROWS; 0: the marching default of make_launch)
                    static const int mixed_rows = env_int("MGX_MIXED_ROWS", 0);
                    Launch gr = make_launch(d.N, 2, d.N - 1, mixed_rows > 0 ? mixed_rows : rpc);
                    if (gr.blocks > s->partial_cap) {
                        const int R = (int)(((long)gr.strips * (d.N - 1) / kWavesPerBlock + s->partial_cap - 9) / (s->partial_cap - 8)) + 1;
                        gr = make_launch(d.N, 2, d.N - 1, R);
                    }
                    hipLaunchKernelGGL(k_update_residual, dim3(gr.blocks), dim3(kBlock), 0, s->stream, (const double*)d.u,
                                       (const float*)w.u, (const double*)d.b, (double*)d.tmp, (float*)w.b, pending_scale,
                                       1.0 / next_scale, s->partial, d.N, d.pitch, w.pitch, 1, d.N, gr.R, gr.strips, gr.chunks);
                    hipLaunchKernelGGL(k_reduce_partials, dim3(1), dim3(kReduceThreads), 0, s->stream, s->partial, gr.blocks, s->sum_dev);
                    std::swap(d.u, d.tmp);
                } else {
                    launch_residual<double, 2>((const double*)d.u, (const double*)d.b, w.b, w.pitch, s->partial,
                                               s->sum_dev, 1.0 / next_scale, d.N, d.pitch, 1, d.N, rpc, s->stream, s->partial_cap, d.rows);
                }
            }
            HIPCHK(s, hipMemcpyAsync(s->sum_host, s->sum_dev, sizeof(double), hipMemcpyDeviceToHost, s->stream));
            HIPCHK(s, hipStreamSynchronize(s->stream));
            hist.push_back(std::sqrt(*s->sum_host));
        }
    }
    HIPCHK(s, hipGetLastError());
    HIPCHK(s, hipStreamSynchronize(s->stream));
    const auto t1 = std::chrono::steady_clock::now();
    if (stats) {
        stats->cycles = k;
        stats->initial_residual = hist.front();
        stats->final_residual = hist.back();
        stats->converged = (hist.back() <= tol * hist.front()) ? 1 : 0;
        stats->seconds = std::chrono::duration<double>(t1 - t0).count();
        stats->fine_updates = s->fine_updates;
        stats->history_len = (int)hist.size();
    }
    if (history)
        for (int i = 0; i < (int)hist.size() && i < history_cap; ++i) history[i] = hist[i];
    return MGX_OK;
}

// ---- measurement ------------------------------------------------------------------------
int mgx_profile_reset(mgx_handle s)
{
    if (!s) return MGX_ERR_INVALID;
    if (s->dist) {
        int rc = dist_prof_collect(s, s->dist);
        if (rc) return rc;
        for (int i = 0; i < MGX_PROF_COUNT; ++i) { s->dist->prof_ms[i] = 0.0; s->dist->prof_launches[i] = 0; s->dist->prof_sweeps[i] = 0; }
        return MGX_OK;
    }
    int rc = prof_collect(s);
    if (rc) return rc;
    for (int i = 0; i < MGX_PROF_COUNT; ++i) { s->prof_ms[i] = 0.0; s->prof_launches[i] = 0; s->prof_sweeps[i] = 0; }
    return MGX_OK;
}

int mgx_profile_get(mgx_handle s, mgx_profile* out)
{
    if (!s || !out) return MGX_ERR_INVALID;
    if (s->dist) {
        int rc = dist_prof_collect(s, s->dist);
        if (rc) return rc;
        for (int i = 0; i < MGX_PROF_COUNT; ++i) {
            out->ms[i] = s->dist->prof_ms[i]; out->launches[i] = s->dist->prof_launches[i]; out->sweeps[i] = s->dist->prof_sweeps[i];
        }
        return MGX_OK;
    }
    int rc = prof_collect(s);
    if (rc) return rc;
    for (int i = 0; i < MGX_PROF_COUNT; ++i) {
        out->ms[i] = s->prof_ms[i];
        out->launches[i] = s->prof_launches[i];
        out->sweeps[i] = s->prof_sweeps[i];
    }
    return MGX_OK;
}

int mgx_time_smoother(mgx_handle s, int sweeps, double* ms)
{
    if (!s || !ms || sweeps < 1) return MGX_ERR_INVALID;
    NO_DIST(s)
    if (int vr = var_ready(s, s->cfg.finest_level, s->cfg.finest_level)) return vr;
    hipEvent_t a, b;
    HIPCHK(s, hipEventCreate(&a));
    HIPCHK(s, hipEventCreate(&b));
    Level& l = s->lv[s->cfg.finest_level];
    HIPCHK(s, hipEventRecord(a, s->stream));
    if (s->var) { if (l.f64) smooth_var_t<double>(s, l, sweeps); else smooth_var_t<float>(s, l, sweeps); }
    else if (l.f64) smooth_t<double>(s, l, sweeps); else smooth_t<float>(s, l, sweeps);
    HIPCHK(s, hipEventRecord(b, s->stream));
    HIPCHK(s, hipEventSynchronize(b));
    float f = 0.f;
    HIPCHK(s, hipEventElapsedTime(&f, a, b));
    (void)hipEventDestroy(a);
    (void)hipEventDestroy(b);
    *ms = f;
    return MGX_OK;
}

// =====================================================================================
// slab-level operators on caller-owned device memory
// =====================================================================================
static int slab_check(const mgx_slab* s)
{
    if (!s || s->level < 2 || s->level > 15 || s->rows < 1) return MGX_ERR_INVALID;
    if (s->dtype != MGX_DTYPE_F32 && s->dtype != MGX_DTYPE_F64) return MGX_ERR_INVALID;
    if (s->arith != MGX_ARITH_SEPARATE && s->arith != MGX_ARITH_FMA) return MGX_ERR_INVALID;
    // the slab must lie inside the grid: rows row0 .. row0 + rows - 1 of rows 0 .. N
    if (s->row0 < 0 || s->row0 + s->rows > (1 << s->level) + 1) return MGX_ERR_INVALID;
    return MGX_OK;
}

long mgx_slab_scratch_doubles(const mgx_slab* s)
{
    if (slab_check(s)) return -1;
    const int N = 1 << s->level;
    const int rpc = env_int("MGX_ROWS", 0);
    long cap = 0;
    for (int r : {rpc, 1}) {
        cap = std::max(cap, sumsq_blocks<double>(N, s->rows, r));
        cap = std::max(cap, sumsq_blocks<float>(N, s->rows, r));
    }
    return cap + 8;
}

static int slab_smooth(int smoother, const mgx_slab* s, void* u, const void* b, void* tmp, int row_lo, int row_hi,
                       int mu, double omega, int shrink, int* result_in_tmp, void* stream)
{
    if (slab_check(s) || !u || !b || !tmp || mu < 0) return MGX_ERR_INVALID;
    const int N = 1 << s->level;
    const long pitch = level_pitch(s->level, s->dtype);
    const int first = 1 - s->row0, last = N - s->row0;      // unknown rows are [first, last)
    const int rpc = env_int("MGX_ROWS", 0);
    FuseCfg fc = fuse_cfg();
    fc.arith = s->arith;
    int parity = 0, rc;
    if (s->dtype == MGX_DTYPE_F64)
        rc = smooth_block<double>(smoother, (double*)u, (const double*)b, (double*)tmp, N, pitch, s->rows, row_lo, row_hi,
                                  mu, omega, shrink != 0, first, last, s->row0 & 1, rpc, fc, (hipStream_t)stream, &parity);
    else
        rc = smooth_block<float>(smoother, (float*)u, (const float*)b, (float*)tmp, N, pitch, s->rows, row_lo, row_hi,
                                 mu, omega, shrink != 0, first, last, s->row0 & 1, rpc, fc, (hipStream_t)stream, &parity);
    if (rc) return rc;
    if (result_in_tmp) *result_in_tmp = parity;
    return hipGetLastError() == hipSuccess ? MGX_OK : MGX_ERR_HIP;
}

int mgx_slab_jacobi(const mgx_slab* s, void* u, const void* b, void* tmp, int row_lo, int row_hi, int mu,
                    double omega, int shrink, int* result_in_tmp, void* stream)
{
    return slab_smooth(MGX_SMOOTHER_JACOBI, s, u, b, tmp, row_lo, row_hi, mu, omega, shrink, result_in_tmp, stream);
}

int mgx_slab_rbgs(const mgx_slab* s, void* u, const void* b, void* tmp, int row_lo, int row_hi, int mu,
                  int shrink, int* result_in_tmp, void* stream)
{
    return slab_smooth(MGX_SMOOTHER_RBGS, s, u, b, tmp, row_lo, row_hi, mu, 1.0, shrink, result_in_tmp, stream);
}

} // extern "C"

namespace {
// mu sweeps on a slab with the cycle's transfers folded into the passes (k_jacobi_cycle on the
// window of rows the slab holds).  Local row numbers in, global ones to the kernel.
template <typename T, int SM, int AR>
int slab_cycle_t(const mgx_slab* f, T* u, const T* b, T* tmp, int row_lo, int row_hi, int mu, double omega,
                        const mgx_slab* c, const T* coarse_e, T* coarse_b, int crow_lo, int crow_hi, int restrict_mode, int zero_in,
                        double* scratch, double* sum_dev, int* result_in_tmp, hipStream_t st)
{
    constexpr bool rbgs = (SM == 1);
    constexpr int per = rbgs ? 2 : 1;
    const int N = 1 << f->level;
    const long pitch = level_pitch(f->level, f->dtype);
    const int first = 1 - f->row0, last = N - f->row0;        // local unknown rows [first, last)
    FuseCfg fc = fuse_cfg();
    fc.arith = AR;
    const int post = coarse_b ? 1 : (sum_dev ? 2 : 0);
    if (coarse_e && coarse_b) {                  // correction and restriction may meet in one pass: at most 8 levels
        fc.fold_kmax = std::min(fc.fold_kmax, 8); fc.fold_kmax_big = std::min(fc.fold_kmax_big, 8);
    }
    int parts[64];
    const int np = plan_folded(fc, rbgs ? MGX_SMOOTHER_RBGS : MGX_SMOOTHER_JACOBI, N, mu, post, sizeof(T) == 8, parts, coarse_e != nullptr);
    const T om = (T)omega;
    const T c0 = (T)(1.0 - (double)om);
    const T c1 = (T)((double)om / 4.0);
    FoldArgs fa;
    fa.restrict_mode = restrict_mode;
    fa.partial = scratch;
    fa.win.row_first = std::max(f->row0, 0);
    fa.win.row_last = std::min(f->row0 + f->rows - 1, N);
    fa.win.crow_first = 0; fa.win.crow_last = -1; fa.win.emit_lo = 0; fa.win.emit_hi = 0;
    if (c) {
        fa.cpitch = level_pitch(c->level, c->dtype);
        fa.win.crow_first = std::max(c->row0, 0);
        fa.win.crow_last = std::min(c->row0 + c->rows - 1, N / 2);
        fa.win.emit_lo = std::max(c->row0 + crow_lo, 1);
        fa.win.emit_hi = std::min(c->row0 + crow_hi, N / 2);
        // base pointers moved back so that GLOBAL coarse rows index them (never dereferenced outside the window)
        if (coarse_e) fa.coarse_e = coarse_e - (long)c->row0 * fa.cpitch;
        if (coarse_b) fa.coarse_b = coarse_b - (long)c->row0 * fa.cpitch;
    }
    const long back = (long)f->row0 * pitch;
    // Small slab ranges (2048^2 slabs of 256 rows at 8 GPUs, the edge bands of an overlapped exchange): a marching pass is
    // latency-bound there - R + 2K row steps one after the other, however few rows - and the register-tile kernel does
    // the whole block in one launch of independent tiles.  Same arithmetic in the same order: same bits.
    {
        const long tile_points = env_int("MGX_SLAB_TILE_POINTS", 1 << 20);          // (read per call: the parity tests switch it)
        const int lo = std::max(std::max(row_lo, first), 1), hi = std::min(std::min(row_hi, last), f->rows - 1);
        if (fc.tile_max_n > 0 && per * mu <= fc.tile_k && hi > lo && (long)(hi - lo) * N <= tile_points) {
            FoldArgs ta = fa;
            ta.row_lo = lo + f->row0; ta.row_hi = hi + f->row0;
            int flips = 0;
            const int nb = smooth_tiled<T, SM, AR>(u - back, b - back, tmp - back, N, pitch, mu, omega, fc.tile_k, ta, coarse_e != nullptr,
                                                   post, zero_in != 0, st, &flips);
            if (nb >= 0) {
                if (post == 2) hipLaunchKernelGGL(k_reduce_partials, dim3(1), dim3(kReduceThreads), 0, st, scratch, nb, sum_dev);
                if (result_in_tmp) *result_in_tmp = flips & 1;
                return hipGetLastError() == hipSuccess ? MGX_OK : MGX_ERR_HIP;
            }
        }
    }
    T* src = u; T* dst = tmp;
    int done = 0, blocks = 0;
    for (int p = 0; p < np; ++p) {
        const int sw = parts[p], K = per * sw;
        const bool P = coarse_e && p == 0;
        const int Q = (p == np - 1) ? post : 0;
        fa.zero_in = (p == 0 && zero_in) ? 1 : 0;              // PS:613: the first pass synthesises the zero guess
        if (fa.zero_in && !rbgs && K == 1) return MGX_ERR_INVALID;      // (a stand-alone single sweep reads its input)
        // rows the later passes still consume; the norm / restriction stage of the last pass also
        // needs the result one / two rows beyond its range (it recomputes those rows itself, from
        // this pass's output)
        const int ext = per * (mu - (done + sw)) + (p != np - 1 ? (post == 2 ? 1 : (post == 1 ? 2 : 0)) : 0);
        // never beyond the unknown rows, and never the slab's first or last row unless it is a global
        // boundary's neighbour: a row is updated from the rows above and below it, and the single-sweep
        // kernel (k_jacobi_rows) reads them without asking whether they exist
        const int lo = std::max(std::max(row_lo - ext, first), 1), hi = std::min(std::min(row_hi + ext, last), f->rows - 1);
        if (hi > lo) {
            const int R = fuse_rows(fc, N, K, sizeof(T) == 8, hi - lo);
            if (P || Q) {
                if (!cycle_k_supported(K, rbgs, sizeof(T) == 8, Q, P, AR)) return MGX_ERR_INVALID;
                fa.row_lo = lo + f->row0; fa.row_hi = hi + f->row0;
                int rc;
                const int Rc = fuse_rows_auto(fc, N, K, sizeof(T) == 8) ? -R : R;
                if (P && Q == 2) rc = launch_cycle<T, 1, 2, SM, AR>(K, src - back, b - back, dst - back, fa, N, pitch, c0, c1, Rc, st);
                else if (P && Q == 1) rc = launch_cycle<T, 1, 1, SM, AR>(K, src - back, b - back, dst - back, fa, N, pitch, c0, c1, Rc, st);
                else if (P) rc = launch_cycle<T, 1, 0, SM, AR>(K, src - back, b - back, dst - back, fa, N, pitch, c0, c1, Rc, st);
                else if (Q == 1) rc = launch_cycle<T, 0, 1, SM, AR>(K, src - back, b - back, dst - back, fa, N, pitch, c0, c1, Rc, st);
                else rc = launch_cycle<T, 0, 2, SM, AR>(K, src - back, b - back, dst - back, fa, N, pitch, c0, c1, Rc, st);
                if (rc < 0) return MGX_ERR_INVALID;
                if (Q == 2) blocks = rc;
            } else if (!rbgs && K == 1) {
                if (launch_jacobi<T>(src, b, dst, N, pitch, lo, hi, omega, env_int("MGX_ROWS", 0), st, f->rows, AR)) return MGX_ERR_INVALID;
            } else if (!launch_fused<T, SM, AR>(K, src, b, dst, N, pitch, lo, hi, c0, c1, first - 1, last, f->row0 & 1, R, st, f->rows, fa.zero_in)) {
                return MGX_ERR_INVALID;
            }
        }
        std::swap(src, dst);
        done += sw;
    }
    if (post == 2) hipLaunchKernelGGL(k_reduce_partials, dim3(1), dim3(kReduceThreads), 0, st, scratch, blocks, sum_dev);
    if (result_in_tmp) *result_in_tmp = np & 1;
    return hipGetLastError() == hipSuccess ? MGX_OK : MGX_ERR_HIP;
}

} // namespace

extern "C" {

int mgx_slab_cycle(const mgx_slab* f, void* u, const void* b, void* tmp, int row_lo, int row_hi, int mu, double omega,
                   int smoother, const mgx_slab* c, const void* coarse_e, void* coarse_b, int crow_lo, int crow_hi,
                   int restrict_mode, int zero_in, double* scratch, double* sum_dev, int* result_in_tmp, void* stream)
{
    if (zero_in && coarse_e) return MGX_ERR_INVALID;
    // the folded restriction wants its range to start on an odd global row (include/mgx.h) - whichever kernel serves the call
    if (coarse_b && !((std::max(row_lo + f->row0, 1)) & 1)) return MGX_ERR_INVALID;
    if (slab_check(f) || !u || !b || !tmp || mu < 1 || mu > 64 || row_hi <= row_lo) return MGX_ERR_INVALID;
    if ((coarse_e || coarse_b) && (slab_check(c) || c->level != f->level - 1 || c->dtype != f->dtype)) return MGX_ERR_INVALID;
    if (coarse_b && sum_dev) return MGX_ERR_INVALID;
    if (sum_dev && !scratch) return MGX_ERR_INVALID;
    if (coarse_b && (crow_lo < 0 || crow_hi > c->rows || crow_hi < crow_lo)) return MGX_ERR_INVALID;
    const int N = 1 << f->level;
    if (row_lo < 0 || row_hi > f->rows || row_lo + f->row0 < 1 || row_hi + f->row0 > N) return MGX_ERR_INVALID;
    hipStream_t st = (hipStream_t)stream;
    const bool rbgs = (smoother == MGX_SMOOTHER_RBGS);
#define MGX_SLAB_CYCLE(T, SM, AR)                                                                                       \
    slab_cycle_t<T, SM, AR>(f, (T*)u, (const T*)b, (T*)tmp, row_lo, row_hi, mu, omega, c, (const T*)coarse_e, (T*)coarse_b, \
                            crow_lo, crow_hi, restrict_mode, zero_in, scratch, sum_dev, result_in_tmp, st)
    const bool fma = f->arith == MGX_ARITH_FMA;
    if (f->dtype == MGX_DTYPE_F64)
        return rbgs ? MGX_SLAB_CYCLE(double, 1, 0) : (fma ? MGX_SLAB_CYCLE(double, 0, 1) : MGX_SLAB_CYCLE(double, 0, 0));
    return rbgs ? MGX_SLAB_CYCLE(float, 1, 0) : (fma ? MGX_SLAB_CYCLE(float, 0, 1) : MGX_SLAB_CYCLE(float, 0, 0));
#undef MGX_SLAB_CYCLE
}

int mgx_slab_restrict(const mgx_slab* f, const void* u, const void* b, const mgx_slab* c, void* cb, void* zero_u,
                      int crow_lo, int crow_hi, int restrict_mode, int fused, void* stream)
{
    if (slab_check(f) || slab_check(c) || !b || !cb || (fused && !u)) return MGX_ERR_INVALID;
    if (c->level != f->level - 1 || c->dtype != f->dtype) return MGX_ERR_INVALID;
    const int N = 1 << f->level;
    const long pitch = level_pitch(f->level, f->dtype), cpitch = level_pitch(c->level, c->dtype);
    const int off = 2 * c->row0 - f->row0;
    // coarse rows must be unknown rows; fine rows 2I+off-2 .. 2I+off+2 must exist (fused), +-1 otherwise
    const int halo = fused ? 2 : 1;
    if (crow_lo < 0 || crow_hi > c->rows || crow_lo + c->row0 < 1 || crow_hi + c->row0 > N / 2) return MGX_ERR_INVALID;
    if (crow_hi > crow_lo && (2 * crow_lo + off - halo < 0 || 2 * (crow_hi - 1) + off + halo > f->rows - 1)) return MGX_ERR_INVALID;
    const int rpc = env_int("MGX_ROWS", 0);
    if (f->dtype == MGX_DTYPE_F64)
        launch_restrict<double>((const double*)u, (const double*)b, (double*)cb, (double*)zero_u, N, pitch, cpitch,
                                crow_lo, crow_hi, off, restrict_mode, fused != 0, rpc, (hipStream_t)stream);
    else
        launch_restrict<float>((const float*)u, (const float*)b, (float*)cb, (float*)zero_u, N, pitch, cpitch,
                               crow_lo, crow_hi, off, restrict_mode, fused != 0, rpc, (hipStream_t)stream);
    return hipGetLastError() == hipSuccess ? MGX_OK : MGX_ERR_HIP;
}

int mgx_slab_prolong(const mgx_slab* f, void* u, const mgx_slab* c, const void* e, int row_lo, int row_hi, int add,
                     void* stream)
{
    if (slab_check(f) || slab_check(c) || !u || !e) return MGX_ERR_INVALID;
    if (c->level != f->level - 1 || c->dtype != f->dtype) return MGX_ERR_INVALID;
    const int N = 1 << f->level;
    const long pitch = level_pitch(f->level, f->dtype), cpitch = level_pitch(c->level, c->dtype);
    const int off = 2 * c->row0 - f->row0;
    if (row_lo < 0 || row_hi > f->rows || row_lo + f->row0 < 1 || row_hi + f->row0 > N) return MGX_ERR_INVALID;
    if (row_hi > row_lo) {
        const int ylo = row_lo - off, yhi = row_hi - 1 - off;
        if (ylo < 0 || (yhi >> 1) + (yhi & 1) > c->rows - 1) return MGX_ERR_INVALID;
    }
    const int rpc = env_int("MGX_ROWS", 0);
    if (f->dtype == MGX_DTYPE_F64)
        launch_prolong<double>((double*)u, (const double*)e, N, pitch, cpitch, row_lo, row_hi, off, add != 0, rpc, (hipStream_t)stream);
    else
        launch_prolong<float>((float*)u, (const float*)e, N, pitch, cpitch, row_lo, row_hi, off, add != 0, rpc, (hipStream_t)stream);
    return hipGetLastError() == hipSuccess ? MGX_OK : MGX_ERR_HIP;
}

int mgx_slab_residual_sumsq(const mgx_slab* s, const void* u, const void* b, int row_lo, int row_hi,
                            double* scratch, double* sum_dev, void* stream)
{
    if (slab_check(s) || !u || !b || !scratch || !sum_dev) return MGX_ERR_INVALID;
    const int N = 1 << s->level;
    const long pitch = level_pitch(s->level, s->dtype);
    if (row_lo < 1 || row_hi > s->rows - 1 || row_lo + s->row0 < 1 || row_hi + s->row0 > N || row_hi <= row_lo) return MGX_ERR_INVALID;
    const int rpc = env_int("MGX_ROWS", 0);
    if (s->dtype == MGX_DTYPE_F64)
        launch_residual<double, 1>((const double*)u, (const double*)b, nullptr, 0, scratch, sum_dev, 1.0, N, pitch, row_lo, row_hi, rpc, (hipStream_t)stream, -1, s->rows);
    else
        launch_residual<float, 1>((const float*)u, (const float*)b, nullptr, 0, scratch, sum_dev, 1.0, N, pitch, row_lo, row_hi, rpc, (hipStream_t)stream, -1, s->rows);
    return hipGetLastError() == hipSuccess ? MGX_OK : MGX_ERR_HIP;
}

// ---- multi-GPU: ranks, plans, helpers ----------------------------------------------------------------
int mgx_create_rank(const mgx_config* cfg, int rank, int world, const void* rccl_id, const mgx_transport* transport,
                    mgx_handle* out)
{
    if (!cfg || !out) { g_create_error = "null argument"; return MGX_ERR_INVALID; }
    *out = nullptr;
    if (cfg->coarsest_level < 2 || cfg->finest_level < cfg->coarsest_level || cfg->finest_level > 15 ||
        cfg->mu1 < 0 || cfg->mu2 < 0 || !(cfg->omega > 0.0 && cfg->omega < 2.0) || cfg->smoother < 0 || cfg->smoother > 1 ||
        cfg->dtype < 0 || cfg->dtype > 2 || cfg->restrict_mode < 0 || cfg->restrict_mode > 1 || cfg->bottom < 0 || cfg->bottom > 1 || cfg->arith < 0 || cfg->arith > 1 ||
        (rank >= 0 && (world < 1 || rank >= world)) || (rank < 0 && cfg->n_gpus < 2)) {
        g_create_error = "invalid configuration";
        return MGX_ERR_INVALID;
    }
    log_runtime_libs("mgx_create_rank");
    mgx_solver* s = new (std::nothrow) mgx_solver();
    if (!s) { g_create_error = "out of host memory"; return MGX_ERR_ALLOC; }
    s->cfg = *cfg;
    const int rc = dist_create(s, &s->cfg, rank, world, rccl_id, transport);
    if (rc != MGX_OK) { g_create_error = s->err; mgx_destroy(s); return rc; }
    *out = s;
    return MGX_OK;
}

// Which ROCm runtime libraries this process has mapped (/proc/self/maps): libmgx is built against /opt/rocm
// (RUNPATH), and a host application that loaded another copy of libamdhip64 / libhsa-runtime64 / librccl
// first (the torch wheel bundles its own, with the same SONAMEs) would run this library's code objects on
// THAT stack.  One path per line; returns the number of bytes written (without the terminator), < 0 on error.
int mgx_runtime_libs(char* buf, size_t cap)
{
    if (!buf || cap == 0) return -1;
    buf[0] = 0;
    FILE* fh = std::fopen("/proc/self/maps", "r");
    if (!fh) return -1;
    std::vector<std::string> seen;
    char line[4096];
    while (std::fgets(line, sizeof line, fh)) {
        const char* path = std::strchr(line, '/');
        if (!path) continue;
        std::string p(path);
        while (!p.empty() && (p.back() == '\n' || p.back() == ' ')) p.pop_back();
        const char* names[] = {"libamdhip64", "libhsa-runtime64", "librccl", "libmgx", "libhiprtc", "libamd_comgr"};
        bool want = false;
        for (const char* n : names) want = want || p.find(n) != std::string::npos;
        if (!want || std::find(seen.begin(), seen.end(), p) != seen.end()) continue;
        seen.push_back(p);
    }
    std::fclose(fh);
    std::string out;
    for (const auto& p : seen) { out += p; out += '\n'; }
    if (out.size() + 1 > cap) out.resize(cap - 1);
    std::memcpy(buf, out.c_str(), out.size() + 1);
    return (int)out.size();
}

int mgx_rccl_unique_id(void* out128)
{
    if (!out128) return MGX_ERR_INVALID;
    ncclUniqueId id;
    if (ncclGetUniqueId(&id) != ncclSuccess) return MGX_ERR_HIP;
    std::memset(out128, 0, 128);
    std::memcpy(out128, &id, sizeof(id) < 128 ? sizeof(id) : 128);
    return MGX_OK;
}

#ifdef MGX_WAVE_TRACE
// debug build only: the wave trace of the last k_jacobi_cycle launch; returns the number of waves
__attribute__((visibility("default"))) int mgx_debug_wave_trace(void* out, int cap)
{
    int n = 0;
    if (hipDeviceSynchronize() != hipSuccess) return -1;
    if (hipMemcpyFromSymbol(&n, HIP_SYMBOL(mgx::g_wave_trace_n), sizeof(int)) != hipSuccess) return -1;
    n = std::min(std::min(n, cap), 1 << 16);
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(mgx::g_wave_trace), (size_t)n * sizeof(mgx::WaveTrace)) != hipSuccess) return -1;
    return n;
}
#endif

long mgx_dist_exchanges(mgx_handle s) { return (s && s->dist) ? s->dist->exchanges : -1; }
long mgx_dist_overlapped(mgx_handle s) { return (s && s->dist) ? s->dist->overlapped : -1; }

int mgx_memcpy_d2h(void* dst_host, const void* src_dev, size_t bytes, void* stream)
{
    if (hipMemcpyAsync(dst_host, src_dev, bytes, hipMemcpyDeviceToHost, (hipStream_t)stream) != hipSuccess) return MGX_ERR_HIP;
    return hipStreamSynchronize((hipStream_t)stream) == hipSuccess ? MGX_OK : MGX_ERR_HIP;
}

int mgx_memcpy_h2d(void* dst_dev, const void* src_host, size_t bytes, void* stream)
{
    if (hipMemcpyAsync(dst_dev, src_host, bytes, hipMemcpyHostToDevice, (hipStream_t)stream) != hipSuccess) return MGX_ERR_HIP;
    return hipStreamSynchronize((hipStream_t)stream) == hipSuccess ? MGX_OK : MGX_ERR_HIP;
}

struct mgx_dist_planner { mgx::DistPlanner p; };

int mgx_plan_create(const mgx_config* cfg, int n_slabs, int g, int fold, int deep, mgx_plan_handle* out)
{
    if (!cfg || !out) { g_plan_error = "null argument"; return MGX_ERR_INVALID; }
    *out = nullptr;
    const int cut = dist_cut_level(*cfg, n_slabs);
    mgx_dist_planner* h = new (std::nothrow) mgx_dist_planner();
    if (!h) return MGX_ERR_ALLOC;
    if (h->p.init(plan_cfg_of(*cfg, n_slabs, g, cut, fold != 0, deep != 0)) != MGX_OK) {
        g_plan_error = h->p.err;
        delete h;
        return MGX_ERR_INVALID;
    }
    *out = h;
    return MGX_OK;
}

int mgx_plan_destroy(mgx_plan_handle p) { delete p; return MGX_OK; }
const char* mgx_plan_last_error(void) { return g_plan_error.c_str(); }
int mgx_plan_cut_level(mgx_plan_handle p) { return p ? p->p.c.cut : -1; }

int mgx_plan_level(mgx_plan_handle p, int level, mgx_dist_level* out)
{
    if (!p || !out || level <= p->p.c.cut || level > p->p.c.finest) return MGX_ERR_INVALID;
    *out = p->p.L(level);
    return MGX_OK;
}

int mgx_plan_cut_share(mgx_plan_handle p, int* row0, int* rows)
{
    if (!p || !row0 || !rows) return MGX_ERR_INVALID;
    *row0 = p->p.c_row0; *rows = p->p.c_rows;
    return MGX_OK;
}

int mgx_plan_guess_set(mgx_plan_handle p, int all_rows)
{
    if (!p) return MGX_ERR_INVALID;
    if (all_rows) p->p.guess_set(); else p->p.guess_changed();
    return MGX_OK;
}

static int plan_emit(mgx_plan_handle p, mgx_dist_op* ops, int cap, bool norm)
{
    if (!p || (!ops && cap > 0)) return MGX_ERR_INVALID;
    std::vector<mgx_dist_op> v;
    // emit on a copy first: a too-small buffer must not advance the planner's halo state
    mgx::DistPlanner trial = p->p;
    if (norm) trial.emit_norm(v); else trial.emit_vcycle(v);
    if ((int)v.size() > cap) return -(int)v.size();
    p->p = trial;
    for (size_t i = 0; i < v.size(); ++i) ops[i] = v[i];
    return (int)v.size();
}
int mgx_plan_vcycle(mgx_plan_handle p, mgx_dist_op* ops, int cap) { return plan_emit(p, ops, cap, false); }
int mgx_plan_norm(mgx_plan_handle p, mgx_dist_op* ops, int cap) { return plan_emit(p, ops, cap, true); }
int mgx_plan_fmg(mgx_plan_handle p, mgx_dist_op* ops, int cap)
{
    if (!p || (!ops && cap > 0)) return MGX_ERR_INVALID;
    std::vector<mgx_dist_op> v;
    mgx::DistPlanner trial = p->p;
    trial.emit_fmg(v);
    if ((int)v.size() > cap) return -(int)v.size();
    p->p = trial;
    for (size_t i = 0; i < v.size(); ++i) ops[i] = v[i];
    return (int)v.size();
}

} // extern "C"
