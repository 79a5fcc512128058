// mgx_launch.hpp - host side of the kernels of mgx_kernels.hpp: launch geometry, the typed
// launch wrappers, the pass planner (how mu sweeps are split into fused / folded launches) and
// the tuning knobs.  No solver state here: the single-GPU handle (mgx.hip) and the slab
// operators of the multi-GPU driver use the same functions.
#pragma once

#include "../../include/mgx.h"
#include "mgx_kernels.hpp"

#include <algorithm>
#include <cstdlib>
#include <vector>

namespace mgx {

inline long level_pitch(int level, int dtype)
{
    const long N = 1L << level;
    const long align = (dtype == MGX_DTYPE_F64) ? 32 : 64;   // 256 bytes
    return (N + 1 + align - 1) / align * align;
}

inline int env_int(const char* name, int dflt)
{
    const char* s = getenv(name);
    return (s && *s) ? atoi(s) : dflt;
}

// temporal fusion knobs: levels per pass, chunk height (0 = by grid size), smallest
// fused grid, levels per pass for the folded kernels
struct FuseCfg {
    int kmax; int rows; int min_n; int fold_kmax; int fold_kmax_big; int tile_max_n; int tile_k; int fold_kmax_nopost;
    // explicit pass plans (sweeps per pass) for the pre- / post-smoothing block of grids with
    // N >= plan_min_n: tuning knobs MGX_PLAN_PRE / MGX_PLAN_POST ("8,2"), MGX_PLAN_MIN_N
    int plan_pre[8] = {0}; int n_pre = 0; int plan_post[8] = {0}; int n_post = 0; int plan_min_n = 8192;
    // mgx_config.arith (MGX_ARITH_*): not a tuning knob - it selects which of the two arithmetic modes of the
    // Jacobi update every smoother kernel uses (jac_pt in mgx_kernels.hpp); set by the handle / the slab, never from the environment
    int arith = 0;
};

// ---- typed operator launches ----------------------------------------------------
// rows_alloc: rows the arrays hold.  A sweep of rows [row_lo,row_hi) reads rows row_lo-1 .. row_hi:
// ranges that would read outside the allocation are refused here, in one place for every caller
// (and the kernels predicate their row loads on rows_alloc as well).
template <typename T>
int launch_jacobi(const T* vin, const T* b, T* vout, int N, long pitch, int row_lo, int row_hi,
                  double omega, int rpc, hipStream_t st, int rows_alloc, int arith = 0)
{
    if (row_hi <= row_lo) return MGX_OK;
    if (row_lo < 1 || row_hi > rows_alloc - 1) return MGX_ERR_INVALID;
    // PS:127, 138-140: the float path evaluates the scalars in double from the
    // float omega and narrows them
    const T om = (T)omega;
    const T c0 = (T)(1.0 - (double)om);
    const T c1 = (T)((double)om / 4.0);
    if (rpc <= 0) {
        // default: one wave per row and strip (see k_jacobi_rows)
        const Launch g = make_launch(N, VecOf<T>::W, row_hi - row_lo, 1);
        if (arith) hipLaunchKernelGGL((k_jacobi_rows<T, 1>), dim3(g.blocks), dim3(kBlock), 0, st, vin, b, vout, N, pitch,
                                      row_lo, row_hi, g.strips, c0, c1, rows_alloc);
        else hipLaunchKernelGGL((k_jacobi_rows<T, 0>), dim3(g.blocks), dim3(kBlock), 0, st, vin, b, vout, N, pitch,
                                row_lo, row_hi, g.strips, c0, c1, rows_alloc);
        return MGX_OK;
    }
    const Launch g = make_launch(N, VecOf<T>::W, row_hi - row_lo, rpc);
    if (arith) hipLaunchKernelGGL((k_jacobi<T, 1>), dim3(g.blocks), dim3(kBlock), 0, st, vin, b, vout, N, pitch,
                                  row_lo, row_hi, g.R, g.strips, g.chunks, c0, c1, rows_alloc);
    else hipLaunchKernelGGL((k_jacobi<T, 0>), dim3(g.blocks), dim3(kBlock), 0, st, vin, b, vout, N, pitch,
                            row_lo, row_hi, g.R, g.strips, g.chunks, c0, c1, rows_alloc);
    return MGX_OK;
}

template <typename T>
void launch_rbgs(const T* vin, const T* b, T* vout, int N, long pitch, int row_lo, int row_hi,
                 int row_parity, int bnd_lo, int bnd_hi, int rpc, hipStream_t st)
{
    if (row_hi <= row_lo) return;
    const Launch g = make_launch(N, VecOf<T>::W, row_hi - row_lo, rpc);
    hipLaunchKernelGGL((k_rbgs<T>), dim3(g.blocks), dim3(kBlock), 0, st, vin, b, vout, N, pitch,
                       row_lo, row_hi, g.R, g.strips, g.chunks, row_parity, bnd_lo, bnd_hi);
}

// chunk height >= R (in steps of `step`) for which a chunk's step count R + extra is a whole number of
// loop trips, or one short of it when parity rules the exact fit out
// K levels in one pass (k_jacobi_fused<T,K,SM>): K Jacobi sweeps (SM = 0) or K/2
// red-black Gauss-Seidel sweeps (SM = 1)
template <typename T, int K, int SM, int AR>
void launch_fused_k(const T* vin, const T* b, T* vout, int N, long pitch, int row_lo, int row_hi,
                    T c0, T c1, int bnd_lo, int bnd_hi, int row_parity, int R, hipStream_t st, int rows_alloc, int zero_in)
{
    constexpr int OUT = fused_out_lanes<K, VecOf<T>::W>();
    R = trip_rows(R, 2 * K, trip_steps<T>(), 1);       // whole loop trips in the interior body
    Launch g = make_launch(N, VecOf<T>::W, row_hi - row_lo, R);
    g.strips = (N / VecOf<T>::W + OUT - 1) / OUT;
    const long waves = (long)g.strips * g.chunks;
    g.blocks = (int)(((waves + kWavesPerBlock - 1) / kWavesPerBlock + 7) / 8 * 8);
    hipLaunchKernelGGL((k_jacobi_fused<T, K, SM, AR>), dim3(g.blocks), dim3(kBlock), 0, st, vin, b, vout, N, pitch,
                       row_lo, row_hi, g.R, g.strips, g.chunks, c0, c1, bnd_lo, bnd_hi, row_parity, rows_alloc, zero_in);
}

// rows_alloc: number of rows the arrays hold (every load is bounded by it)
template <typename T, int SM, int AR>
bool launch_fused(int K, const T* vin, const T* b, T* vout, int N, long pitch, int row_lo, int row_hi,
                  T c0, T c1, int bnd_lo, int bnd_hi, int row_parity, int R, hipStream_t st, int rows_alloc, int zero_in)
{
    switch (K) {
        case 2: launch_fused_k<T, 2, SM, AR>(vin, b, vout, N, pitch, row_lo, row_hi, c0, c1, bnd_lo, bnd_hi, row_parity, R, st, rows_alloc, zero_in); return true;
        case 4: launch_fused_k<T, 4, SM, AR>(vin, b, vout, N, pitch, row_lo, row_hi, c0, c1, bnd_lo, bnd_hi, row_parity, R, st, rows_alloc, zero_in); return true;
        case 6: launch_fused_k<T, 6, SM, AR>(vin, b, vout, N, pitch, row_lo, row_hi, c0, c1, bnd_lo, bnd_hi, row_parity, R, st, rows_alloc, zero_in); return true;
        case 8: launch_fused_k<T, 8, SM, AR>(vin, b, vout, N, pitch, row_lo, row_hi, c0, c1, bnd_lo, bnd_hi, row_parity, R, st, rows_alloc, zero_in); return true;
        case 10: launch_fused_k<T, 10, SM, AR>(vin, b, vout, N, pitch, row_lo, row_hi, c0, c1, bnd_lo, bnd_hi, row_parity, R, st, rows_alloc, zero_in); return true;
        default: break;
    }
    if constexpr (SM == 0) {
        switch (K) {
            case 3: launch_fused_k<T, 3, 0, AR>(vin, b, vout, N, pitch, row_lo, row_hi, c0, c1, bnd_lo, bnd_hi, row_parity, R, st, rows_alloc, zero_in); return true;
            case 5: launch_fused_k<T, 5, 0, AR>(vin, b, vout, N, pitch, row_lo, row_hi, c0, c1, bnd_lo, bnd_hi, row_parity, R, st, rows_alloc, zero_in); return true;
            default: break;
        }
    }
    return false;
}

// chunk height of a fused pass: a chunk recomputes 2K halo rows, so the deeper the pass the
// taller the chunk, against the parallelism short chunks give (all measured, bench.py sweeps
// of MGX_FUSE_ROWS): K <= 2: 8 rows (flat from 8 to 24); K = 3, 4: 16-24 rows (8192^2 RB-GS
// V(2,2) 1.52 -> 1.32 ms, Jacobi V(4,3) 1.45 -> 1.27 ms against 8 rows; 4-7 % at 4096^2 and
// 2048^2); K >= 5: N/128 clamped to [8, 64] in double (flat between 48 and 96 at 8192^2), to
// [8, 32] in float (a float wave covers twice the columns, so a grid has half the strips:
// 8192^2 fp32 V(10,10) finest level 0.96 -> 0.90 ms at 32 rows)
// deep double passes (K >= 8: two workgroups of four waves per CU = 2048 waves per round): whole
// rounds.  One wave per (chunk, strip of 52-54 vectors): with cpr = 2048 / strips chunk rows per round,
// `rows` rows in m rounds take R = rows / (m cpr); m = the fewest rounds with R <= 200.  Measured (one
// pass, us): 4096^2 whole 183 at 36 rows, 154 at 84 (m = 1, 490 workgroups), 167-190 at 48-72 and
// 90-108; 16384^2 whole 1947 at 72, 1807 at 200; 8192^2 whole 487-498 at 72 and 484-492 at 168 (flat; on
// a box that clocks down under the f64 load, 540 against 492).  Fewer, taller chunks also recompute
// fewer halo rows (2K + 3 per chunk).
inline int fuse_rows_deep(int N, int rows)
{
    const int strips = (N / 2 + 51) / 52;
    const int cpr = std::max(1, 2048 / strips);
    for (int m = 1; m <= 64; ++m) {
        const int R = (rows + m * cpr - 1) / (m * cpr);
        if (R <= 200) return std::max(R, 16);
    }
    return 64;
}

// deep double passes on big grids, height not given by MGX_FUSE_ROWS: launch_cycle_k sizes the chunks itself
// (edge tiles shorter than interior ones) - it is handed -fuse_rows(...)
// (float: the 10-level passes, whose rhs window is in LDS like the deep double ones - until round 3 they ran 32-row
// chunks, 52 row steps for 32 rows)
inline bool fuse_rows_auto(const FuseCfg& fc, int N, int K, bool f64)
{
    static const bool f32_auto = env_int("MGX_F32_AUTO_ROWS", 1) != 0;
    return fc.rows <= 0 && (f64 ? K >= 8 : (K >= 10 && f32_auto)) && N >= 2048;
}

inline int shallow_big_n() { static const int n = env_int("MGX_SHALLOW_BIG_N", 8192); return n; }
// (4096^2: 48 rows - config 2's V(2,1) cycle 0.416-0.418 -> 0.389-0.395 ms, the 4096^2 level of config 3 -4 %; 96 there: 0.402)
inline int shallow_mid_rows() { static const int r = env_int("MGX_SHALLOW_ROWS_MID", 48); return r; }

inline int fuse_rows(const FuseCfg& fc, int N, int K, bool f64 = true, int rows = 0)
{
    if (fc.rows > 0) return fc.rows;
    if (K <= 2) {
        // (8192^2 and up: 96 rows like the K <= 4 passes below - together −2..−3.5 % on the finest-level part of a V(2,1) cycle)
        static const int big2 = env_int("MGX_SHALLOW2_ROWS_BIG", 96);
        return N >= shallow_big_n() ? big2 : (N >= 4096 ? shallow_mid_rows() : 8);
    }
    if (K <= 4) {
        // (8192^2 and up: 96 rows - finest-level part of the red-black V(2,1) cycle 0.828-0.833 -> 0.789 ms, Jacobi V(2,1)
        // 0.811-0.817 -> 0.786-0.794; round 1's 24 rows paid (24 + 2K) / 24 in recomputed rows)
        static const int big = env_int("MGX_SHALLOW_ROWS_BIG", 96);
        return N >= shallow_big_n() ? big : (N >= 4096 ? shallow_mid_rows() : 16);
    }
    if (f64 && K >= 8 && N >= 2048) return fuse_rows_deep(N, rows > 0 ? rows : N - 1);
    int R = N / 128;
    if (R < 8) R = 8;
    if (R > (f64 ? 64 : 32)) R = f64 ? 64 : 32;
    return R;
}

inline FuseCfg fuse_cfg()
{
    FuseCfg f;
    f.kmax = env_int("MGX_FUSE", 10);          // levels per pass; 1 disables temporal fusion
    if (f.kmax < 1) f.kmax = 1;
    if (f.kmax > 10) f.kmax = 10;
    f.rows = env_int("MGX_FUSE_ROWS", 0);      // 0: chosen from the grid size
    if (f.rows < 0) f.rows = 0;
    // smallest grid (N = 2^L) on which fused / folded passes replace single sweeps
    f.min_n = std::max(64, env_int("MGX_FUSE_MIN_N", 256));
    // Levels per pass for the folded kernels.  They carry one more level window and the
    // transfer state, so their sweet spot is shallower than the plain fused kernel's and
    // flat: measured in one process on one MI355X, V(10,10) at 8192^2 fp64 takes 2.63 ms as
    // [5,5] and 2.62 as [10] (244-256 VGPRs, 2 waves/SIMD); [5,5] is better on smaller grids.
    f.fold_kmax = std::max(1, std::min(f.kmax, env_int("MGX_FOLD_KMAX", 10)));
    // The same for grids with N >= 8192 (separate knob): there the deep variant [10] is
    // device-dependent - 1.55 vs 1.62 ms for the finest level on one MI355X, 1.87 vs 1.50 ms on
    // another (VALU-bound passes follow the clock the chip holds; the HBM-bound [5,5] does not).
    f.fold_kmax_big = std::max(1, std::min(f.kmax, env_int("MGX_FOLD_KMAX_BIG", 10)));
    // Whole levels up to this N (= 2^L) are smoothed by the LDS tile kernel, all sweeps of a
    // block (up to tile_k levels) per launch; 0 disables it.
    f.tile_max_n = std::max(0, env_int("MGX_TILE_MAX_N", 1024));
    f.tile_k = std::max(2, std::min(10, env_int("MGX_TILE_K", 10)));
    // levels per folded pass for blocks that end WITHOUT a residual stage (post-smoothing below
    // the finest level): those passes keep c1 * b in their window and are cheaper per level
    f.fold_kmax_nopost = std::max(1, std::min(f.kmax, env_int("MGX_FOLD_KMAX_NOPOST", 10)));
    auto parse = [](const char* name, int* out) {
        const char* v = std::getenv(name);
        int n = 0;
        while (v && *v && n < 8) {
            char* end = nullptr;
            const long k = std::strtol(v, &end, 10);
            if (end == v || k < 1 || k > 10) return 0;
            out[n++] = (int)k;
            v = (*end == ',') ? end + 1 : end;
            if (*end && *end != ',') return 0;
        }
        return n;
    };
    f.n_pre = parse("MGX_PLAN_PRE", f.plan_pre);
    f.n_post = parse("MGX_PLAN_POST", f.plan_post);
    f.plan_min_n = env_int("MGX_PLAN_MIN_N", 8192);
    return f;
}

// Per-sweep throughput of a fused launch relative to one stand-alone sweep,
// measured on MI355X at 8192^2 (tools/microbench, profiles/r01_fused_microbench.md).
// Index = sweeps per launch; 0 = not instantiated.  Jacobi: K = 5 is poor in
// float because it needs a second halo lane per side for one extra column.
// Re-measured after the fused Jacobi passes started keeping c1 * b in their rhs window
// (K - 1 fewer multiplications per point: fp64 K=8 1138 -> 1349 G upd/s, fp32 K=6 1457 -> 1852,
// fp32 K=10 1371 -> 2019).
constexpr double kFuseRate64[11] = {0, 1.00, 1.80, 2.45, 3.16, 3.81, 4.80, 0, 5.60, 0, 5.63};
// float again after the row operators were written on pairs (all arithmetic packed:
// v_pk_add_f32 / v_pk_mul_f32): K=5 1508 -> 1743, K=8 1731 -> 2301 G upd/s.
constexpr double kFuseRate32[11] = {0, 1.00, 1.67, 2.38, 3.10, 3.77, 4.26, 0, 4.98, 0, 4.67};
// red-black Gauss-Seidel: s sweeps = 2 s levels, s <= 5
constexpr double kFuseRateGS64[11] = {0, 1.00, 1.81, 2.50, 3.19, 3.12, 0, 0, 0, 0, 0};
constexpr double kFuseRateGS32[11] = {0, 1.00, 1.79, 2.20, 2.87, 2.69, 0, 0, 0, 0, 0};

// split mu sweeps into fused launches minimising the modelled time; parts[] gets
// the sweeps of each launch, returns their count.  kmax bounds the LEVELS per pass.
inline int plan_fusion(int mu, int kmax, bool f64, int* parts, bool rbgs = false)
{
    const double* rate = rbgs ? (f64 ? kFuseRateGS64 : kFuseRateGS32) : (f64 ? kFuseRate64 : kFuseRate32);
    const int per = rbgs ? 2 : 1;
    const int smax = std::max(1, std::min(kmax, 10) / per);
    std::vector<double> best(mu + 1, 1e300);
    std::vector<int> pick(mu + 1, 1);
    best[0] = 0.0;
    for (int m = 1; m <= mu; ++m)
        for (int k = 1; k <= std::min(m, smax); ++k) {
            if (rate[k] <= 0.0) continue;
            // + a small per-launch cost so that equal-rate splits prefer fewer launches
            const double c = best[m - k] + (double)k / rate[k] + 0.02;
            if (c < best[m]) { best[m] = c; pick[m] = k; }
        }
    int n = 0;
    for (int m = mu; m > 0; m -= pick[m]) parts[n++] = pick[m];
    return n;
}

// mu smoother sweeps on rows [row_lo,row_hi) ping-ponging a <-> b2; *parity = 1 when
// the result ends in `b2`.  Unknown rows are [first, last) (so the global
// boundary rows are first-1 and last).  shrink: deep-halo mode, the sweeps still
// to come after a launch widen its range by `per` rows each at every interior
// edge.  rows_alloc bounds every row that is read (validated, never assumed).
template <typename T>
int smooth_block(int smoother, T* a, const T* rhs, T* b2, int N, long pitch, int rows_alloc, int row_lo, int row_hi,
                 int mu, double omega, bool shrink, int first, int last, int row_parity, int rpc, const FuseCfg& fc,
                 hipStream_t st, int* parity, int* launches = nullptr)
{
    const bool rbgs = (smoother == MGX_SMOOTHER_RBGS);
    const int per = rbgs ? 2 : 1;
    const T om = (T)omega;
    const T c0 = (T)(1.0 - (double)om);
    const T c1 = (T)((double)om / 4.0);
    T* src = a; T* dst = b2;
    int flips = 0;
    int done = 0;
    // Fused launches pay (R + 2K)/R redundant rows and need enough chunks to fill
    // the chip: measured worthwhile from 256^2 up, with R growing with the grid.
    const bool allow_fuse = fc.kmax > per && N >= fc.min_n && (row_hi - row_lo) >= 64 && mu <= 64;
    std::vector<int> parts(mu > 0 ? mu : 1, 1);
    const int nparts = allow_fuse ? plan_fusion(mu, fc.kmax, sizeof(T) == 8, parts.data(), rbgs) : mu;
    const int bl = first - 1, bh = last;
    for (int p = 0; p < nparts; ++p) {
        const int sw = allow_fuse ? parts[p] : 1;               // sweeps in this launch
        const int K = per * sw;                                  // levels in this launch
        const int ext = shrink ? per * (mu - (done + sw)) : 0;   // rows the later launches still consume
        const int lo = std::max(row_lo - ext, first), hi = std::min(row_hi + ext, last);
        if (hi > lo) {
            // rows read: [lo-K, hi+K) clipped to the global boundary rows
            const int rd_lo = std::max(lo - K, bl), rd_hi = std::min(hi + K - 1, bh);
            if (rd_lo < 0 || rd_hi > rows_alloc - 1) return MGX_ERR_INVALID;
            if (!rbgs && sw == 1) {
                const int rc = launch_jacobi<T>(src, rhs, dst, N, pitch, lo, hi, omega, rpc, st, rows_alloc, fc.arith);
                if (rc) return rc;
            } else if (rbgs && !allow_fuse) {
                launch_rbgs<T>(src, rhs, dst, N, pitch, lo, hi, row_parity, bl, bh, rpc, st);
            } else {
                const int R = fuse_rows(fc, N, K, sizeof(T) == 8, hi - lo);
                const bool ok = rbgs ? launch_fused<T, 1, 0>(K, src, rhs, dst, N, pitch, lo, hi, c0, c1, bl, bh, row_parity, R, st, rows_alloc, 0)
                                : (fc.arith ? launch_fused<T, 0, 1>(K, src, rhs, dst, N, pitch, lo, hi, c0, c1, bl, bh, row_parity, R, st, rows_alloc, 0)
                                            : launch_fused<T, 0, 0>(K, src, rhs, dst, N, pitch, lo, hi, c0, c1, bl, bh, row_parity, R, st, rows_alloc, 0));
                if (!ok) return MGX_ERR_INVALID;
            }
        }
        std::swap(src, dst);
        ++flips;
        done += sw;
    }
    *parity = flips & 1;
    if (launches) *launches = flips;
    return MGX_OK;
}

// ---- smoother passes with the cycle's transfers folded in (k_jacobi_cycle) ----------
struct FoldArgs {
    const void* coarse_e = nullptr;   // PRE: correction to add while loading
    void* coarse_b = nullptr;         // POST 1: restricted residual
    void* coarse_zero = nullptr;      // POST 1: coarse guess to zero
    int restrict_mode = 0;
    double* partial = nullptr;        // POST 2: per-block sums of r^2
    long cpitch = 0;
    int zero_in = 0;                  // the input iterate is all zero: the pass does not read it
    // rows to update, GLOBAL numbers, and the window of rows that exist; row_hi == 0: the whole
    // grid (rows 1..N-1, window 0..N).  Slabs pass base pointers moved back by row0 rows.
    int row_lo = 0, row_hi = 0;
    CycleWin win{0, 0, 0, 0, 0, 0};
};

// chunk geometry of a k_jacobi_cycle launch: mgx_geom.hpp (cycle_geom_pick / cycle_tile_at); the knobs from the environment
// (read at every launch of a deep pass: the parity tests switch them inside one process)
inline GeomKnobs geom_knobs()
{
    GeomKnobs k;
    k.edge_short = env_int("MGX_EDGE_SHORT", 1) != 0;
    k.edge_pct = env_int("MGX_EDGE_PCT", 23);
    k.last_pct = env_int("MGX_LAST_PCT", 38);
    k.min_chunk = std::max(8, env_int("MGX_MIN_CHUNK", 16));
    k.min_rounds = std::max(1, env_int("MGX_MIN_ROUNDS", 1));
    k.min_rounds_rows = env_int("MGX_MIN_ROUNDS_ROWS", 1024);
    k.pair = env_int("MGX_PAIR", 1) != 0;
    k.pair_ratio = std::max(100, env_int("MGX_PAIR_RATIO", 130));
    k.pair_max_rows = env_int("MGX_PAIR_MAX_ROWS", 640);
    k.pair_min_rows = env_int("MGX_PAIR_MIN_ROWS", 150);
    return k;
}
// R < 0: choose the chunk height here (deep double passes: whole rounds of 2048 waves, see fuse_rows_deep;
// -R is the height the uniform rule gave)
template <typename T, int K, int PRE, int POST, int SM, int AR>
int launch_cycle_k(const T* vin, const T* b, T* vout, const FoldArgs& fa, int N, long pitch, T c0, T c1, int R,
                   hipStream_t st)
{
    constexpr int OUT = cycle_out_lanes<K, POST, VecOf<T>::W>();
    constexpr bool BL = cycle_b_in_lds<T, K, POST, SM>();
    constexpr int E = (POST == 1 ? 3 : (POST == 2 ? 2 : 0)) + (cycle_skew<T, K, PRE, POST, SM, AR>() > 0 ? 1 : 0);   // row steps of a chunk beyond R + 2K
    constexpr int kTripSteps = BL ? kBRing : trip_steps<T>();
    const bool whole = (fa.row_hi == 0);
    const int row_lo = whole ? 1 : fa.row_lo, row_hi = whole ? N : fa.row_hi;
    const CycleWin win = whole ? CycleWin{0, N, 0, N / 2, 1, N / 2} : fa.win;
    if (POST == 1 && !(row_lo & 1)) return -1;         // chunks must start on odd rows (POST = 1)
    const int strips = (N / VecOf<T>::W + OUT - 1) / OUT;
    const bool auto_rows = R < 0;
    if (auto_rows) R = -R;
    // the bodies run whole trips (kBRing steps for the deep variants, kTrip otherwise): a chunk is R + 2K + (stage rows)
    // steps long; cycle_geom_pick takes even heights that make it a multiple of the trip (or one short of it).
    // Deep passes (rhs ring in LDS): shorter edge-class tiles; auto_rows: the fewest rounds of resident workgroups with
    // chunks of at most ~200 rows, or ONE round with paired heights (mgx_geom.hpp)
    // an end of the range whose cone (K + 1 rows above, K + 1..2 below: the kernel's `interior`) leaves the unknown rows
    // that exist runs the edge body and gets shorter chunks; the ends of a middle slab, whose halo rows hold the cone, do not
    constexpr int ETOP = POST ? 1 : 0, EBOT = POST == 1 ? 2 : (POST == 2 ? 1 : 0);
    const bool ends_interior = env_int("MGX_SLAB_ENDS_INTERIOR", 1) != 0;      // 0: shortened chunks at every end of every range (the geometry before round 3's last change)
    const bool top_edge = !ends_interior || (row_lo - K - ETOP) < std::max(win.row_first, 1) || (PRE && ((row_lo - K - ETOP) >> 1) < win.crow_first);
    const bool bot_edge = !ends_interior || (row_hi + K + EBOT - 1) > std::min(win.row_last, N - 1) || (PRE && ((row_hi + K + EBOT) >> 1) > win.crow_last);
    GeomKnobs kn = BL ? geom_knobs() : GeomKnobs();
    // (float passes: a wave covers twice the columns, a grid has half the strips and twice the chunks per strip - the paired
    // form pays from shorter chunks: mixed V(10,10) at 8192^2, float finest-level passes 0.403 -> 0.393-0.395 ms at 100 rows)
    if (sizeof(T) == 4 && BL) kn.pair_min_rows = std::min(kn.pair_min_rows, env_int("MGX_PAIR_MIN_ROWS_F32", 100));
    const CycleGeom g = cycle_geom_pick(row_lo, row_hi, strips, 2 * K + E, kTripSteps, R, auto_rows, BL, kn, top_edge, bot_edge);
    const int blocks = g.blocks;
    const T w = (fa.restrict_mode == MGX_RESTRICT_FW16) ? (T)0.0625 : (T)0.25;
    hipLaunchKernelGGL((k_jacobi_cycle<T, K, PRE, POST, SM, AR>), dim3(blocks), dim3(kBlock), 0, st, vin, b, vout,
                       (const T*)fa.coarse_e, (T*)fa.coarse_b, (T*)fa.coarse_zero, w, fa.partial, N, pitch, fa.cpitch,
                       row_lo, row_hi, g.R, strips, g.chunks, g.Re, g.chunks_e, g.row_last0, g.Rl, g.RB, g.n_tall, g.n_short, g.Rf, c0, c1, fa.zero_in, win);
    return blocks;
}

template <typename T, int PRE, int POST, int SM, int AR>
int launch_cycle(int K, const T* vin, const T* b, T* vout, const FoldArgs& fa, int N, long pitch, T c0, T c1, int R,
                 hipStream_t st)
{
    switch (K) {
        case 2: return launch_cycle_k<T, 2, PRE, POST, SM, AR>(vin, b, vout, fa, N, pitch, c0, c1, R, st);
        case 4: return launch_cycle_k<T, 4, PRE, POST, SM, AR>(vin, b, vout, fa, N, pitch, c0, c1, R, st);
        case 6: return launch_cycle_k<T, 6, PRE, POST, SM, AR>(vin, b, vout, fa, N, pitch, c0, c1, R, st);
        default: break;
    }
    // 8 levels: not for float passes that start from the input and end with a residual stage (with the
    // branch-free interior bodies their predicated body extracts an odd-indexed pair of a float4
    // through a stack slot: scratch)
    if constexpr (sizeof(T) == 8 || PRE == 1 || POST == 0) {
        if (K == 8) return launch_cycle_k<T, 8, PRE, POST, SM, AR>(vin, b, vout, fa, N, pitch, c0, c1, R, st);
    }
    // 10 levels with folded stages: not the most register-hungry combinations (correction AND restriction in
    // one pass; red-black GS with the restriction; in float - packed arithmetic wants aligned register pairs -
    // the correction with the norm): with the rhs window in LDS every instantiated variant fits 256
    // registers = two waves per SIMD, float ones included since round 3 (cycle_b_in_lds)
    // (and the float pre-smoothing pass with the restriction only in FMA mode: the separately rounded one spills 3 dwords)
    if constexpr (!(PRE == 1 && POST == 1) && !(SM == 1 && POST == 1) && !(sizeof(T) == 4 && PRE == 1 && POST == 2) &&
                  !(sizeof(T) == 4 && POST == 1 && AR == 0)) {
        if (K == 10) return launch_cycle_k<T, 10, PRE, POST, SM, AR>(vin, b, vout, fa, N, pitch, c0, c1, R, st);
    }
    if constexpr (SM == 0) {
        switch (K) {
            case 1: return launch_cycle_k<T, 1, PRE, POST, 0, AR>(vin, b, vout, fa, N, pitch, c0, c1, R, st);
            case 3: return launch_cycle_k<T, 3, PRE, POST, 0, AR>(vin, b, vout, fa, N, pitch, c0, c1, R, st);
            case 5: return launch_cycle_k<T, 5, PRE, POST, 0, AR>(vin, b, vout, fa, N, pitch, c0, c1, R, st);
            default: break;
        }
    }
    return -1;
}

// levels per pass the folded kernels are instantiated for
inline bool cycle_k_supported(int K, bool rbgs, bool f64, int post, bool pre, int arith)
{
    if (K == 10) return !(pre && post == 1) && !(rbgs && post == 1) && !(!f64 && pre && post == 2) && !(!f64 && post == 1 && arith == 0);
    if (K == 8 && !f64 && !pre && post != 0) return false;
    return rbgs ? (K == 2 || K == 4 || K == 6 || K == 8) : (K >= 1 && K <= 8 && K != 7);
}

// ---- small levels: every sweep of a block in one launch on register tiles (k_tile_smooth) ----
template <typename T, int SM, int PRE, int POST, int AR>
int launch_tile(const T* vin, const T* b, T* vout, const FoldArgs& fa, int N, long pitch, T c0, T c1, int levels,
                hipStream_t st)
{
    const int He = levels + tile_extra<POST>();
    const int TH = kTileSY - 2 * He, TW = kTileSX - 2 * He;
    if (TH < 8 || TW < 8) return -1;
    // whole grid (fa.row_hi == 0), or rows [row_lo, row_hi) of a slab whose base pointers were moved back by row0 rows
    const bool whole = (fa.row_hi == 0);
    const int row_lo = whole ? 1 : fa.row_lo, row_hi = whole ? N : fa.row_hi;
    const CycleWin win = whole ? CycleWin{0, N, 0, N / 2, 1, N / 2} : fa.win;
    const int tiles_y = (row_hi - row_lo + TH - 1) / TH, tiles_x = (N - 1 + TW - 1) / TW;
    if (tiles_y < 1) return 0;
    const T w = (fa.restrict_mode == MGX_RESTRICT_FW16) ? (T)0.0625 : (T)0.25;
    hipLaunchKernelGGL((k_tile_smooth<T, SM, PRE, POST, AR>), dim3(tiles_y * tiles_x), dim3(kBlock), 0, st, vin, b, vout,
                       (const T*)fa.coarse_e, (T*)fa.coarse_b, (T*)fa.coarse_zero, w, fa.partial, N, pitch, fa.cpitch,
                       levels, c0, c1, tiles_x, fa.zero_in, row_lo, row_hi, win);
    return tiles_y * tiles_x;
}

// mu sweeps of a whole level, a <-> b2 ping-pong (*flips launches made); pre / post as in
// smooth_folded_t.  Returns the number of norm partials (post == 2), < 0 on a launch error.
template <typename T, int SM, int AR>
int smooth_tiled(T* a, const T* rhs, T* b2, int N, long pitch, int mu, double omega, int tile_k, FoldArgs fa,
                 bool pre, int post, bool zero_in, hipStream_t st, int* flips)
{
    constexpr int per = (SM == 1) ? 2 : 1;
    const T om = (T)omega;
    const T c0 = (T)(1.0 - (double)om);
    const T c1 = (T)((double)om / 4.0);
    const int smax = std::max(1, tile_k / per);            // sweeps per launch
    const int np = (mu + smax - 1) / smax;
    T* src = a; T* dst = b2;
    int blocks = 0;
    for (int p = 0; p < np; ++p) {
        const int sw = mu / np + (p < mu % np ? 1 : 0);
        const bool P = pre && p == 0;
        const int Q = (p == np - 1) ? post : 0;
        fa.zero_in = (p == 0 && zero_in) ? 1 : 0;
        int rc;
        if (P && Q == 2) rc = launch_tile<T, SM, 1, 2, AR>(src, rhs, dst, fa, N, pitch, c0, c1, per * sw, st);
        else if (P && Q == 1) rc = launch_tile<T, SM, 1, 1, AR>(src, rhs, dst, fa, N, pitch, c0, c1, per * sw, st);
        else if (P) rc = launch_tile<T, SM, 1, 0, AR>(src, rhs, dst, fa, N, pitch, c0, c1, per * sw, st);
        else if (Q == 1) rc = launch_tile<T, SM, 0, 1, AR>(src, rhs, dst, fa, N, pitch, c0, c1, per * sw, st);
        else if (Q == 2) rc = launch_tile<T, SM, 0, 2, AR>(src, rhs, dst, fa, N, pitch, c0, c1, per * sw, st);
        else rc = launch_tile<T, SM, 0, 0, AR>(src, rhs, dst, fa, N, pitch, c0, c1, per * sw, st);
        if (rc < 0) return -1;
        if (Q == 2) blocks = rc;
        std::swap(src, dst);
    }
    *flips = np;
    return blocks;
}

template <typename T>
void launch_restrict(const T* v, const T* b, T* cb, T* czero, int N, long pitch, long cpitch,
                     int crow_lo, int crow_hi, int fine_row_off, int mode, bool fused, int rpc, hipStream_t st)
{
    if (crow_hi <= crow_lo) return;
    Launch g = make_launch(N, VecOf<T>::W, crow_hi - crow_lo, rpc > 0 ? (rpc + 1) / 2 : 0);
    const T w = (mode == MGX_RESTRICT_FW16) ? (T)0.0625 : (T)0.25;
    if (fused)
        hipLaunchKernelGGL((k_restrict<T, true>), dim3(g.blocks), dim3(kBlock), 0, st, v, b, cb, czero, N, pitch,
                           cpitch, crow_lo, crow_hi, fine_row_off, g.R, g.strips, g.chunks, w);
    else
        hipLaunchKernelGGL((k_restrict<T, false>), dim3(g.blocks), dim3(kBlock), 0, st, v, b, cb, czero, N, pitch,
                           cpitch, crow_lo, crow_hi, fine_row_off, g.R, g.strips, g.chunks, w);
}

template <typename T>
void launch_prolong(T* v, const T* e, int N, long pitch, long cpitch, int row_lo, int row_hi,
                    int fine_row_off, bool add, int rpc, hipStream_t st)
{
    if (row_hi <= row_lo) return;
    const Launch g = make_launch(N, VecOf<T>::W, row_hi - row_lo, rpc);
    if (add)
        hipLaunchKernelGGL((k_prolong<T, true>), dim3(g.blocks), dim3(kBlock), 0, st, v, e, N, pitch, cpitch,
                           row_lo, row_hi, fine_row_off, g.R, g.strips, g.chunks);
    else
        hipLaunchKernelGGL((k_prolong<T, false>), dim3(g.blocks), dim3(kBlock), 0, st, v, e, N, pitch, cpitch,
                           row_lo, row_hi, fine_row_off, g.R, g.strips, g.chunks);
}

// blocks needed by the sum-of-squares kernel for a given geometry
template <typename T> long sumsq_blocks(int N, int rows, int rpc)
{
    return make_launch(N, VecOf<T>::W, rows, rpc).blocks;
}

// sum (b - A u)^2 over rows -> sum_dev[0]; MODE 2 also writes scaled float residual
// rows_alloc: rows the arrays hold (rows row_lo-1 .. row_hi are read; the kernel predicates on it)
template <typename T, int MODE>
void launch_residual(const T* v, const T* b, void* out, long pitch_out, double* partial, double* sum_dev,
                     double inv_scale, int N, long pitch, int row_lo, int row_hi, int rpc, hipStream_t st,
                     long partial_cap, int rows_alloc)
{
    Launch g = make_launch(N, VecOf<T>::W, row_hi - row_lo, rpc);
    if (MODE != 0 && partial_cap >= 0 && g.blocks > partial_cap) {
        // never write past the partial-sum buffer: fall back to taller chunks
        const int R = (int)(((long)g.strips * (row_hi - row_lo) / kWavesPerBlock + partial_cap - 9) / (partial_cap - 8)) + 1;
        g = make_launch(N, VecOf<T>::W, row_hi - row_lo, R);
    }
    hipLaunchKernelGGL((k_residual<T, MODE>), dim3(g.blocks), dim3(kBlock), 0, st, v, b, out, pitch_out, partial,
                       inv_scale, N, pitch, row_lo, row_hi, g.R, g.strips, g.chunks, rows_alloc);
    if (MODE != 0)
        hipLaunchKernelGGL(k_reduce_partials, dim3(1), dim3(kReduceThreads), 0, st, partial, g.blocks, sum_dev);
}


// ---- one translation unit per group of kernels --------------------------------------------------------
// The smoother kernels are instantiated for every (type, depth, folded stages, smoother, arithmetic mode):
// a few hundred device functions, three minutes of compile time in one translation unit.  mgx.hip therefore
// only DECLARES the launch wrappers' instantiations; mgx_inst.hip, compiled once per group (Makefile:
// -DMGX_INST_KIND / _T / _SMAR / _PP), defines them.  (SM, AR) pairs: Jacobi separate / Jacobi FMA / red-black GS.
#define MGX_FOR_SMAR(X, T) X(T, 0, 0) X(T, 0, 1) X(T, 1, 0)
#define MGX_DECL_CYCLE_PP(T, PRE, POST, SM, AR) \
    extern template int launch_cycle<T, PRE, POST, SM, AR>(int, const T*, const T*, T*, const FoldArgs&, int, long, T, T, int, hipStream_t);
#define MGX_DECL_CYCLE(T, SM, AR) MGX_DECL_CYCLE_PP(T, 1, 2, SM, AR) MGX_DECL_CYCLE_PP(T, 1, 0, SM, AR) MGX_DECL_CYCLE_PP(T, 0, 1, SM, AR) \
    MGX_DECL_CYCLE_PP(T, 0, 2, SM, AR) MGX_DECL_CYCLE_PP(T, 1, 1, SM, AR)
#define MGX_DECL_FUSED(T, SM, AR) \
    extern template bool launch_fused<T, SM, AR>(int, const T*, const T*, T*, int, long, int, int, T, T, int, int, int, int, hipStream_t, int, int);
#define MGX_DECL_TILE(T, SM, AR) \
    extern template int smooth_tiled<T, SM, AR>(T*, const T*, T*, int, long, int, double, int, FoldArgs, bool, int, bool, hipStream_t, int*);
#if !defined(MGX_INST_KIND) && !defined(MGX_SINGLE_TU)
MGX_FOR_SMAR(MGX_DECL_CYCLE, double) MGX_FOR_SMAR(MGX_DECL_CYCLE, float)
MGX_FOR_SMAR(MGX_DECL_FUSED, double) MGX_FOR_SMAR(MGX_DECL_FUSED, float)
MGX_FOR_SMAR(MGX_DECL_TILE, double) MGX_FOR_SMAR(MGX_DECL_TILE, float)
#endif

} // namespace mgx
