// mgx_kernels.hpp — hand-written gfx950 (CDNA4, wave64) stencil kernels for the
// 2-D Poisson multigrid hot path.  Device code only; launch geometry at the end.
//
// What each kernel replaces in the reference (PS = Poissons_SYCL.cpp,
// MF = Multigrid_functions.cpp):
//   k_jacobi_rows / k_jacobi   PS:137-145  one sweep (gemv + scal + scal + add + add; K1-K5)
//   k_jacobi_fused<T,K,SM>     K levels per pass: K Jacobi sweeps (SM 0) or K/2 red-black
//                              Gauss-Seidel sweeps (SM 1) - temporal fusion of PS:137-145
//   k_jacobi_cycle<..PRE,POST> the same pass with the cycle's transfers folded in:
//                              PS:620-624 correction on load, PS:604-613 residual +
//                              restriction + zero guess, or the residual norm
//   k_rbgs                     one red-black GS sweep (absent from the reference)
//   k_residual                 PS:604-607  (2 gemv + add + sub; K6-K9), also sum r^2 (D10)
//   k_restrict                 PS:531-546  restriction2d (fused with the residual when FUSED)
//   k_prolong                  PS:337-425  interpolation2d, + PS:623 correction add (K10)
//
// Storage (DESIGN.md "Data layout in HBM"): one level = the full node grid,
// rows 0..N and columns 0..N with N = 2^L, *including* the zero Dirichlet ring
// the reference eliminates (PS:188-198, 224).  Row pitch is a multiple of 256 B
// so every row starts on a cache line; columns N+1..pitch-1 are zero padding.
// A thread owns one 16-byte vector (2 doubles / 4 floats) of a row, a wave owns
// 64 consecutive vectors, and every global access is a 16-byte access at a
// 16-byte-aligned address.  The x-neighbours come from the adjacent lanes by DPP
// wavefront shifts, and the outermost lanes of a wave are halo lanes (they load
// and compute but never store), so no wave ever depends on another wave's
// registers, LDS or output.  Multi-level kernels march down a chunk of rows
// keeping rolling row windows in registers, so each value is read from HBM once
// per pass whatever the number of sweeps the pass performs.
//
// Row bounds: the smoothers, the residual kernel and the fused / folded passes take the
// allocation height (rows_alloc, or the slab's row window) and predicate every row they read
// or write on it; k_restrict and k_prolong rely on the range checks of their launch wrappers'
// callers (mgx_slab_restrict / mgx_slab_prolong validate every row they will touch).
#pragma once

#include <hip/hip_runtime.h>
#include "mgx_geom.hpp"
#include <stdint.h>
#include <type_traits>

namespace mgx {

// compile-time loop: f(std::integral_constant<int, I>) for I in [B, E)
template <int B, int E, typename F> __device__ __forceinline__ void static_for(F&& f)
{
    if constexpr (B < E) {
        f(std::integral_constant<int, B>{});
        static_for<B + 1, E>(f);
    }
}

constexpr int kWave = 64;
constexpr int kBlock = 256;                 // 4 waves
constexpr int kWavesPerBlock = kBlock / kWave;
constexpr int kOutLanes = kWave - 2;        // lanes 1..62 store; 0 and 63 are halo lanes

template <typename T> struct VecOf;
template <> struct VecOf<double> { using type = double2; static constexpr int W = 2; };
template <> struct VecOf<float>  { using type = float4;  static constexpr int W = 4; };

// ---- tiny vector helpers ----------------------------------------------------
__device__ __forceinline__ double2 vzero(double2*) { return make_double2(0.0, 0.0); }
__device__ __forceinline__ float4  vzero(float4*)  { return make_float4(0.f, 0.f, 0.f, 0.f); }

template <typename V> __device__ __forceinline__ V vload(const void* p, bool pred)
{
    V z = vzero((V*)nullptr);
    if (pred) z = *reinterpret_cast<const V*>(p);
    return z;
}
template <typename V> __device__ __forceinline__ void vstore(void* p, const V& v, bool pred)
{
    if (pred) *reinterpret_cast<V*>(p) = v;
}

__device__ __forceinline__ double first(const double2& v) { return v.x; }
__device__ __forceinline__ double last(const double2& v)  { return v.y; }
__device__ __forceinline__ float  first(const float4& v)  { return v.x; }
__device__ __forceinline__ float  last(const float4& v)   { return v.w; }

// value held by the lane one to the left / right: DPP wavefront shifts
// (v_mov_b32_dpp wave_shr:1 / wave_shl:1), i.e. plain VALU moves with no LDS
// round trip - __shfl_up/down lower to ds_bpermute_b32, whose latency sits in
// the serial level-to-level dependency chain of the fused kernels.  Semantics:
// __shfl_up/down(x, 1, 64) except on the edge lanes, which get 0 (lane 0's "left"
// and lane 63's "right" are don't-cares: halo lanes never store).
// bound_ctrl = 1: the lane with no source (lane 0 of wave_shr, lane 63 of wave_shl) gets 0 and
// the destination needs no previous value - with `old = x` the compiler had to copy x into the
// destination first (one extra v_mov_b32 per shifted dword: 20 of the 131 vector instructions
// of a K = 5 row step in double).
__device__ __forceinline__ int dpp_shr1(int x) { return __builtin_amdgcn_update_dpp(0, x, 0x138, 0xf, 0xf, true); }
__device__ __forceinline__ int dpp_shl1(int x) { return __builtin_amdgcn_update_dpp(0, x, 0x130, 0xf, 0xf, true); }
__device__ __forceinline__ float from_left(float x)  { return __int_as_float(dpp_shr1(__float_as_int(x))); }
__device__ __forceinline__ float from_right(float x) { return __int_as_float(dpp_shl1(__float_as_int(x))); }
__device__ __forceinline__ double from_left(double x)
{
    return __hiloint2double(dpp_shr1(__double2hiint(x)), dpp_shr1(__double2loint(x)));
}
__device__ __forceinline__ double from_right(double x)
{
    return __hiloint2double(dpp_shl1(__double2hiint(x)), dpp_shl1(__double2loint(x)));
}

// ---- branch-free stores for the interior bodies --------------------------------------------------
// A conditional store (halo lanes do not store; rows outside the chunk do not either) compiles to an
// exec-mask branch, a conditional load to a uniform branch, and every branch ends a basic block: the
// compiler's s_waitcnt insertion then assumes the worst at each join and put `vmcnt(3)` - "all but
// the loads issued in this very step" - in front of every row step: ONE row in flight however many
// the code prefetched (rocprofv3: waves parked on s_waitcnt 40-65 % of their time).  The interior
// bodies therefore contain no branch at all: stores go through a raw buffer descriptor and a lane
// or row that must not store gets an offset beyond the descriptor's range (the hardware drops it),
// conditional accumulations are selects, and "the input is all zero" is a template parameter.
#ifndef MGX_SOFF_LOADS
#define MGX_SOFF_LOADS 1    // interior bodies load through the buffer descriptors with a scalar row offset
#endif
typedef int v4i32 __attribute__((ext_vector_type(4)));
typedef int v2i32 __attribute__((ext_vector_type(2)));
constexpr unsigned kOobOffset = 0xFFFFFF00u;       // beyond every descriptor here (arrays < 4 GiB - 256 B)
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* p, unsigned bytes)
{
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)bytes, 0x00020000);
}
// (the register images are built component by component: a bit_cast of the HIP vector structs went
// through a stack slot - scratch - in the float kernels)
__device__ __forceinline__ void bstore(const double2& v, __amdgpu_buffer_rsrc_t r, unsigned voff)
{
    const v4i32 t = {__double2loint(v.x), __double2hiint(v.x), __double2loint(v.y), __double2hiint(v.y)};
    __builtin_amdgcn_raw_buffer_store_b128(t, r, voff, 0, 0);
}
__device__ __forceinline__ void bstore(const float4& v, __amdgpu_buffer_rsrc_t r, unsigned voff)
{
    const v4i32 t = {__float_as_int(v.x), __float_as_int(v.y), __float_as_int(v.z), __float_as_int(v.w)};
    __builtin_amdgcn_raw_buffer_store_b128(t, r, voff, 0, 0);
}
__device__ __forceinline__ void bstore8(double v, __amdgpu_buffer_rsrc_t r, unsigned voff)
{
    const v2i32 t = {__double2loint(v), __double2hiint(v)};
    __builtin_amdgcn_raw_buffer_store_b64(t, r, voff, 0, 0);
}
__device__ __forceinline__ void bstore8(float2 v, __amdgpu_buffer_rsrc_t r, unsigned voff)
{
    const v2i32 t = {__float_as_int(v.x), __float_as_int(v.y)};
    __builtin_amdgcn_raw_buffer_store_b64(t, r, voff, 0, 0);
}
// ---- branch-free loads for the edge bodies --------------------------------------------------------
// Waves whose dependency cone touches a boundary row or column, or the end of the rows that exist,
// used to run a predicated body (a conditional load or store per predicate, one basic block each):
// twice the instructions of the interior body and one row in flight, and since nothing else runs
// once the interior waves are done, a pass ended with a tail as long as a whole chunk (small levels,
// where every workgroup is resident at once, simply took as long as their slowest edge wave).  The
// edge bodies are now the same straight-line code: their loads go through raw buffer descriptors as
// well - a row or lane that must not be read gets the out-of-range offset and reads 0 - and the
// Dirichlet rows and columns are re-zeroed with selects (+4 v_cndmask per level).
__device__ __forceinline__ double2 bload(double2*, __amdgpu_buffer_rsrc_t r, unsigned voff)
{
    const v4i32 t = __builtin_amdgcn_raw_buffer_load_b128(r, voff, 0, 0);
    return make_double2(__hiloint2double(t.y, t.x), __hiloint2double(t.w, t.z));
}
__device__ __forceinline__ float4 bload(float4*, __amdgpu_buffer_rsrc_t r, unsigned voff)
{
    const v4i32 t = __builtin_amdgcn_raw_buffer_load_b128(r, voff, 0, 0);
    return make_float4(__int_as_float(t.x), __int_as_float(t.y), __int_as_float(t.z), __int_as_float(t.w));
}
// the same with the row part of the address as a SCALAR offset (the instruction's soffset operand): the interior bodies
// address row y of a lane's column as descriptor + lane offset (one VGPR, constant) + (y - rb) * pitch (one s_mul_i32)
// instead of a 64-bit pointer sum per load (five scalar multiplies / adds and a 64-bit vector add each)
__device__ __forceinline__ double2 bload_s(double2*, __amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff)
{
    const v4i32 t = __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0);
    return make_double2(__hiloint2double(t.y, t.x), __hiloint2double(t.w, t.z));
}
__device__ __forceinline__ float4 bload_s(float4*, __amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff)
{
    const v4i32 t = __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0);
    return make_float4(__int_as_float(t.x), __int_as_float(t.y), __int_as_float(t.z), __int_as_float(t.w));
}
__device__ __forceinline__ double bload1_s(double*, __amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff)
{
    const v2i32 t = __builtin_amdgcn_raw_buffer_load_b64(r, voff, soff, 0);
    return __hiloint2double(t.y, t.x);
}
__device__ __forceinline__ float bload1_s(float*, __amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff)
{
    return __int_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 0));
}
__device__ __forceinline__ double bload1(double*, __amdgpu_buffer_rsrc_t r, unsigned voff)
{
    const v2i32 t = __builtin_amdgcn_raw_buffer_load_b64(r, voff, 0, 0);
    return __hiloint2double(t.y, t.x);
}
__device__ __forceinline__ float bload1(float*, __amdgpu_buffer_rsrc_t r, unsigned voff)
{
    return __int_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, voff, 0, 0));
}

// What a body needs to load / store without branches (built once per wave from uniform values).
// Every descriptor starts at the first row its wave can touch (rb fine, crb coarse) and offsets are
// relative to it, so arrays of any size work with 32-bit offsets.
struct FastOut {
    __amdgpu_buffer_rsrc_t out, cb, cz;    // fine output; coarse rhs and coarse guess (POST 1; cz empty when not wanted)
    __amdgpu_buffer_rsrc_t in, rhs, ce;    // edge bodies: input iterate, right-hand side, coarse correction (PRE)
    unsigned lane_off;                     // byte offset of this lane's vector in a fine row
    unsigned clane_off;                    // byte offset of this lane's first coarse column in a coarse row
    unsigned pitch_bytes, cpitch_bytes;
    int rb, crb;                           // the rows the descriptors start at
};
// bytes from row `first` to the end of row `last` of an array, as a descriptor range (a wave reaches
// a few hundred rows beyond its first one: clamping keeps kOobOffset out of range)
__device__ __forceinline__ unsigned rsrc_bytes(int first, int last, unsigned long row_bytes)
{
    if (last < first) return 0u;
    const unsigned long b = (unsigned long)(last - first + 1) * row_bytes;
    return b < 0x7FFFFF00ul ? (unsigned)b : 0x7FFFFF00u;
}
// A uniform value the optimizer cannot see through.  The edge bodies test every level's row against
// the Dirichlet rows; row y-j of step y is row (y+1)-(j+1) of the next step, and with the trip fully
// unrolled the compiler computes each row's lane masks once and keeps them alive for K steps: 40+
// live SGPR pairs, ~150 spilled to VGPR lanes, the VGPRs in turn to scratch.  Recomputing a mask per
// level costs a few scalar instructions.
__device__ __forceinline__ int opaque_s(int x) { asm volatile("" : "+s"(x)); return x; }
// column masks of a lane: its first column is the Dirichlet column 0 / its columns lie at or beyond column N
struct ColMask { bool first, all; };
__device__ __forceinline__ ColMask col_mask(long col, int N) { return ColMask{col == 0, col >= N}; }
// zero the columns that are not unknowns, and everything when the row is not an unknown row (selects)
__device__ __forceinline__ void mask_sel(double2& v, const ColMask& m, bool row_zero)
{
    v.x = (m.first || m.all || row_zero) ? 0.0 : v.x;
    v.y = (m.all || row_zero) ? 0.0 : v.y;
}
__device__ __forceinline__ void mask_sel(float4& v, const ColMask& m, bool row_zero)
{
    v.x = (m.first || m.all || row_zero) ? 0.f : v.x;
    v.y = (m.all || row_zero) ? 0.f : v.y;
    v.z = (m.all || row_zero) ? 0.f : v.z;
    v.w = (m.all || row_zero) ? 0.f : v.w;
}

// ---- wave -> (row chunk, column strip) --------------------------------------
// Blocks are dealt round-robin over the 8 XCDs (b and b+8 share an XCD and its
// L2).  The remap gives each XCD a contiguous range of tiles, x fastest, so the
// halo rows and halo vectors a wave re-reads were fetched by a neighbour on the
// same L2.  gridDim.x is always a multiple of 8 (launch wrappers round up).
struct Tile { int chunk, strip; bool active; };

__device__ __forceinline__ Tile wave_tile(int strips, int chunks)
{
    const int per_xcd = gridDim.x >> 3;
    const int xcd = blockIdx.x & 7;
    // the upper four XCDs walk their ranges backwards: the last chunk row of the grid - boundary waves,
    // which run the slower edge body - is then the first thing XCD 7 starts, not the last
    const int nth = blockIdx.x >> 3;
    const int b = xcd * per_xcd + (xcd >= 4 ? per_xcd - 1 - nth : nth);
    // the wave index as a scalar: everything derived from it (chunk, rows, row offsets, the
    // row predicates) then lives in SGPRs and costs no vector instruction
    const long g = (long)b * kWavesPerBlock + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    Tile t;
    t.chunk = (int)(g / strips);
    t.strip = (int)(g - (long)t.chunk * strips);
    t.active = t.chunk < chunks;
    return t;
}

// k_jacobi_cycle's map.  A workgroup's four waves are four neighbouring tiles.  The waves of the first
// and last strip of a chunk row (the grid's first and last columns) always run the edge body, ~1.25 x an
// interior row step; with the plain map (x fastest) one workgroup in twenty had one such wave.  Here the
// tiles of strips 1 .. S-2 come first (x fastest, contiguous per XCD as in wave_tile) and the two edge
// strips after them, so edge waves share workgroups with edge waves: 8192^2 pass 0.479 -> 0.430 ms,
// 4096^2 0.158 -> 0.146, same chunk height, same kernels (A/B inside one GPU call).  Why it pays that much
// is not established: one-wave workgroups - no workgroup waits for four free wave slots - changed nothing
// on top of it.
// The edge tiles are also a little SHORTER (Re < R rows: first / last chunk of every strip and all chunks
// of the two edge strips), so that in a launch of one round of workgroups the slower waves end with the
// others; measured worth 1-2 % (MGX_EDGE_SHORT=0 makes every tile R rows high):
//   strips 1 .. S-2: `chunks` chunks: the first Re rows, then R rows each up to row_last0, and the last TWO
//                    Rl rows each, [row_last0, row_last0 + Rl) and [row_last0 + Rl, row_hi) - anchored at
//                    the END of the range.  The wave trace (tools/wave_trace.py) showed a pass of one round
//                    ending 30-60 % after its median wave: the waves whose cone or read-ahead reaches the
//                    last row (the last chunk row AND the one before it) take 1.35-1.6 us per row step
//                    against 0.8-1.1 for interior ones, so those two chunk rows are the shortest;
//   strips 0 and S-1: `chunks_e` chunks of Re rows;
//   row_last0 = 0: every tile of every strip R = Re rows high, from the top.
#ifdef MGX_WAVE_TRACE
// debug build only (make TRACE=1): when every wave of the last k_jacobi_cycle launch started and ended
// (100 MHz wall clock), where it ran and what it did - read back with mgx_debug_wave_trace
struct WaveTrace { long long t0, t1; int strip, r0, r1, hw; };
__device__ WaveTrace g_wave_trace[1 << 16];
__device__ int g_wave_trace_n;
#endif
// (the tile map itself is plain C++ shared with the launcher and the CPU tests: mgx_geom.hpp)
__device__ __forceinline__ CTile cycle_tile(int strips, int chunks, int chunks_e, int R, int Re, int row_lo, int row_hi,
                                            int row_last0, int Rl, int RB, int n_tall, int n_short, int Rf)
{
    // the wave index as a scalar: everything derived from it (chunk, rows, row offsets, the row predicates) then
    // lives in SGPRs and costs no vector instruction
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    return cycle_tile_at((int)blockIdx.x, wave, (int)gridDim.x, strips, chunks, chunks_e, R, Re, row_lo, row_hi, row_last0, Rl,
                         RB, n_tall, n_short, Rf);
}

struct Cols {           // per-lane column bookkeeping, shared by all kernels
    int vx;             // vector index of this lane in the row
    long col;           // first column of the vector
    bool ld;            // vector lies inside the row allocation
    bool st;            // this lane stores (not a halo lane, holds interior columns)
};

template <int W> __device__ __forceinline__ Cols lane_cols(int strip, int N, long pitch)
{
    const int lane = threadIdx.x & 63;
    Cols c;
    c.vx = strip * kOutLanes - 1 + lane;
    c.col = (long)c.vx * W;
    c.ld = (c.vx >= 0) && (c.col + W <= pitch);
    c.st = (lane >= 1) && (lane <= kOutLanes) && (c.vx < N / W);
    return c;
}

// off-diagonal sum in the reference's CSR column order N, W, E, S (PS:138)
template <typename T> __device__ __forceinline__ T nbr(T n, T w, T e, T s) { return ((n + w) + e) + s; }

// =============================================================================
// weighted Jacobi, one sweep, out of place:  vout = (1-w) v + (w/4) b + (w/4) S v
//   arithmetic order of PS:138-142: t = c0*v + c1*b ; out = t + c1*(N+W+E+S).
// Algorithmic HBM bytes per updated point: read v + read b + write v' = 3 sizeof(T).
// Rows [row_lo,row_hi) are updated; rows row_lo-1 and row_hi are read only.
// =============================================================================
// fp32 rows are processed as two pairs so that every operation is a packed
// v_pk_add_f32 / v_pk_mul_f32 (IEEE per component: same bits as the scalar form, and the
// off-diagonal sum keeps the order ((N + W) + E) + S).  With cur = (x, y, z, w), l / r the
// neighbours' w / x:  W = (l, x | y, z),  E = (y, z | w, r).
typedef float f32x2 __attribute__((ext_vector_type(2)));
struct NbrPairs { f32x2 t0, t1; };
__device__ __forceinline__ NbrPairs nbr_pairs(const float4& up, const float4& cur, const float4& dn)
{
    const float l = from_left(cur.w), r = from_right(cur.x);
    const f32x2 n0 = {up.x, up.y}, n1 = {up.z, up.w}, s0 = {dn.x, dn.y}, s1 = {dn.z, dn.w};
    const f32x2 A = {l, cur.x}, B = {cur.y, cur.z}, C = {cur.w, r};
    NbrPairs p;
    p.t0 = ((n0 + A) + B) + s0;
    p.t1 = ((n1 + B) + C) + s1;
    return p;
}

// One point of the sweep from cb = c1 * b (rounded once) and nb = ((N + W) + E) + S.
//   AR = 0 (MGX_ARITH_SEPARATE, the default): every IEEE operation of PS:138-142 on its own, in the
//           reference's order - t = c0 v + cb ; v' = t + c1 nb - bit-identical to the CPU oracle's default mode;
//   AR = 1 (MGX_ARITH_FMA): the same expression with its two multiply-adds contracted,
//           v' = fma(c1, nb, fma(c0, v, cb)) - two roundings fewer and 5 instead of 7 vector instructions per
//           point (the deep passes are bound by the vector ALU); bit-identical to the oracle's
//           FMA mode, and within north_star's 1e-10 of the default mode's residual histories (tests).
template <int AR> __device__ __forceinline__ double jac_pt(double c0, double cur, double cb, double c1, double nb)
{
    if constexpr (AR != 0) return __builtin_fma(c1, nb, __builtin_fma(c0, cur, cb));
    else return (c0 * cur + cb) + c1 * nb;
}
template <int AR> __device__ __forceinline__ float jac_pt(float c0, float cur, float cb, float c1, float nb)
{
    if constexpr (AR != 0) return __builtin_fmaf(c1, nb, __builtin_fmaf(c0, cur, cb));
    else return (c0 * cur + cb) + c1 * nb;
}
template <int AR> __device__ __forceinline__ f32x2 jac_pt(float c0, f32x2 cur, f32x2 cb, float c1, f32x2 nb)
{
    if constexpr (AR != 0) {
        const f32x2 C0 = {c0, c0}, C1 = {c1, c1};
        return __builtin_elementwise_fma(C1, nb, __builtin_elementwise_fma(C0, cur, cb));
    } else {
        return (c0 * cur + cb) + c1 * nb;
    }
}

// the sweep with the rhs already multiplied: cb = c1 * b.  The fused kernels use a rhs row at K
// levels; multiplying it when it is loaded saves K - 1 multiplications per point.
template <int AR = 0>
__device__ __forceinline__ double2 jacobi_vec_pre(const double2& up, const double2& cur, const double2& dn,
                                                  const double2& cb, double c0, double c1)
{
    const double l = from_left(cur.y), r = from_right(cur.x);
    double2 o;
    o.x = jac_pt<AR>(c0, cur.x, cb.x, c1, nbr(up.x, l, cur.y, dn.x));
    o.y = jac_pt<AR>(c0, cur.y, cb.y, c1, nbr(up.y, cur.x, r, dn.y));
    return o;
}
template <int AR = 0>
__device__ __forceinline__ float4 jacobi_vec_pre(const float4& up, const float4& cur, const float4& dn,
                                                 const float4& cb, float c0, float c1)
{
    const NbrPairs t = nbr_pairs(up, cur, dn);
    const f32x2 p0 = {cur.x, cur.y}, p1 = {cur.z, cur.w}, b0 = {cb.x, cb.y}, b1 = {cb.z, cb.w};
    const f32x2 o0 = jac_pt<AR>(c0, p0, b0, c1, t.t0);
    const f32x2 o1 = jac_pt<AR>(c0, p1, b1, c1, t.t1);
    return make_float4(o0.x, o0.y, o1.x, o1.y);
}
__device__ __forceinline__ double2 vscale(double c, const double2& v) { return make_double2(c * v.x, c * v.y); }
__device__ __forceinline__ float4 vscale(float c, const float4& v)
{
    const f32x2 a = {v.x, v.y}, b = {v.z, v.w};
    const f32x2 x = c * a, y = c * b;
    return make_float4(x.x, x.y, y.x, y.y);
}
// from the rhs itself (the product c1 * b is rounded once either way: same bits)
template <typename T, int AR = 0>
__device__ __forceinline__ typename VecOf<T>::type
jacobi_vec(const typename VecOf<T>::type& up, const typename VecOf<T>::type& cur,
           const typename VecOf<T>::type& dn, const typename VecOf<T>::type& bb, T c0, T c1)
{
    return jacobi_vec_pre<AR>(up, cur, dn, vscale(c1, bb), c0, c1);
}

template <typename T, int AR = 0>
__global__ void __launch_bounds__(kBlock)
k_jacobi(const T* __restrict__ vin, const T* __restrict__ rhs, T* __restrict__ vout,
         int N, long pitch, int row_lo, int row_hi, int R, int strips, int chunks, T c0, T c1, int rows_alloc)
{
    using V = typename VecOf<T>::type;
    constexpr int W = VecOf<T>::W;
    const Tile t = wave_tile(strips, chunks);
    if (!t.active) return;
    const Cols c = lane_cols<W>(t.strip, N, pitch);
    const int r0 = row_lo + t.chunk * R;
    const int r1 = min(r0 + R, row_hi);
    const T* pv = vin + c.col;
    const T* pb = rhs + c.col;
    T* po = vout + c.col;

    // a row is dereferenced only if the arrays hold it (0 <= row < rows_alloc): the predicates are
    // wave-uniform scalar compares and cost nothing, and a caller's range can no longer fault
    auto has = [&](int y) { return c.ld && y >= 0 && y < rows_alloc; };
    V up = vload<V>(pv + (long)(r0 - 1) * pitch, has(r0 - 1));
    V cur = vload<V>(pv + (long)r0 * pitch, has(r0));
    V dn = vload<V>(pv + (long)(r0 + 1) * pitch, has(r0 + 1));
    V bb = vload<V>(pb + (long)r0 * pitch, has(r0));
    for (int r = r0; r < r1; ++r) {
        // prefetch the next row before computing this one (r+2 <= row_hi+1 is
        // never dereferenced past row_hi: predicate on r + 1 < r1)
        const bool more = (r + 1 < r1);
        const V dn2 = vload<V>(pv + (long)(r + 2) * pitch, has(r + 2) && more);
        const V bb2 = vload<V>(pb + (long)(r + 1) * pitch, has(r + 1) && more);
        V o = jacobi_vec<T, AR>(up, cur, dn, bb, c0, c1);
        if (c.vx == 0) o.x = (T)0;          // column 0 is the Dirichlet boundary
        vstore<V>(po + (long)r * pitch, o, c.st && r >= 0 && r < rows_alloc);
        up = cur; cur = dn; dn = dn2; bb = bb2;
    }
}

// Same sweep with no marching: one wave per (row, strip).  Every wave issues its
// four loads at once and retires; the three reads of each v row (as north, centre
// and south) are served by L2 because vertically adjacent waves are dispatched
// back to back on the same XCD.  Measured faster than the marching form on
// MI355X at every size (tools/microbench, DESIGN.md "Kernel choices").
template <typename T, int AR = 0>
__global__ void __launch_bounds__(kBlock)
k_jacobi_rows(const T* __restrict__ vin, const T* __restrict__ rhs, T* __restrict__ vout,
              int N, long pitch, int row_lo, int row_hi, int strips, T c0, T c1, int rows_alloc)
{
    using V = typename VecOf<T>::type;
    constexpr int W = VecOf<T>::W;
    const Tile t = wave_tile(strips, row_hi - row_lo);
    if (!t.active) return;
    const Cols c = lane_cols<W>(t.strip, N, pitch);
    const int r = row_lo + t.chunk;
    const T* pv = vin + c.col + (long)r * pitch;
    // neighbour rows only where the arrays hold them (wave-uniform: r is a scalar)
    const bool in = r >= 0 && r < rows_alloc;
    const V up = vload<V>(pv - pitch, c.ld && r >= 1 && r <= rows_alloc);
    const V cur = vload<V>(pv, c.ld && in);
    const V dn = vload<V>(pv + pitch, c.ld && r >= -1 && r + 1 < rows_alloc);
    const V bb = vload<V>(rhs + c.col + (long)r * pitch, c.ld && in);
    V o = jacobi_vec<T, AR>(up, cur, dn, bb, c0, c1);
    if (c.vx == 0) o.x = (T)0;
    vstore<V>(vout + c.col + (long)r * pitch, o, c.st && in);
}

// =============================================================================
// red-black Gauss-Seidel, one full sweep (red then black), out of place in one
// pass: vout = GS_black(GS_red(vin)).  Colour = parity of (global row + col),
// red = even.  A wave keeps a 4-row window: to finish row r (black) it needs
// the red-updated rows r-1, r, r+1, and red-updating row r+1 needs the old row
// r+2.  Halo rows and halo lanes recompute the red update redundantly, so no
// wave ever depends on another wave's output (no intra-launch hand-off).
//   point update: v = 0.25 * (b + ((N + W) + E) + S)
// row_parity: parity of the global index of local row 0 (slabs).
// bnd_lo / bnd_hi: local indices of the global boundary rows 0 and N; rows at
// or beyond them are never updated (stay 0) and never dereferenced beyond.
// =============================================================================
template <typename T, int COLOUR>
__device__ __forceinline__ typename VecOf<T>::type
gs_colour_vec(const typename VecOf<T>::type& up, const typename VecOf<T>::type& cur,
              const typename VecOf<T>::type& dn, const typename VecOf<T>::type& bb, int par0);

// par0 = parity of (global row + first column of the vector); the element e of
// the vector is updated when ((par0 + e) & 1) == COLOUR.
template <typename T, int COLOUR>
__device__ __forceinline__ T gs_pick(T oldv, T newv, int par) { return ((par & 1) == COLOUR) ? newv : oldv; }

template <int COLOUR>
__device__ __forceinline__ double2 gs_colour_d(const double2& up, const double2& cur, const double2& dn,
                                               const double2& bb, int par0)
{
    const double l = from_left(cur.y), r = from_right(cur.x);
    double2 o;
    o.x = gs_pick<double, COLOUR>(cur.x, 0.25 * (bb.x + nbr(up.x, l, cur.y, dn.x)), par0);
    o.y = gs_pick<double, COLOUR>(cur.y, 0.25 * (bb.y + nbr(up.y, cur.x, r, dn.y)), par0 + 1);
    return o;
}
template <int COLOUR>
__device__ __forceinline__ float4 gs_colour_f(const float4& up, const float4& cur, const float4& dn,
                                              const float4& bb, int par0)
{
    const NbrPairs t = nbr_pairs(up, cur, dn);
    const f32x2 b0 = {bb.x, bb.y}, b1 = {bb.z, bb.w};
    const f32x2 c0 = 0.25f * (b0 + t.t0), c1 = 0.25f * (b1 + t.t1);
    float4 o;
    o.x = gs_pick<float, COLOUR>(cur.x, c0.x, par0);
    o.y = gs_pick<float, COLOUR>(cur.y, c0.y, par0 + 1);
    o.z = gs_pick<float, COLOUR>(cur.z, c1.x, par0);
    o.w = gs_pick<float, COLOUR>(cur.w, c1.y, par0 + 1);
    return o;
}
template <int COLOUR> __device__ __forceinline__ double2
gs_colour(const double2& u, const double2& c, const double2& d, const double2& b, int p) { return gs_colour_d<COLOUR>(u, c, d, b, p); }
template <int COLOUR> __device__ __forceinline__ float4
gs_colour(const float4& u, const float4& c, const float4& d, const float4& b, int p) { return gs_colour_f<COLOUR>(u, c, d, b, p); }

// zero the columns that are not unknowns (column 0, and columns >= N)
__device__ __forceinline__ void mask_cols(double2& v, long col, int N)
{
    if (col == 0) v.x = 0.0;
    if (col >= N) { v.x = 0.0; v.y = 0.0; }
}
__device__ __forceinline__ void mask_cols(float4& v, long col, int N)
{
    if (col == 0) v.x = 0.f;
    if (col >= N) { v.x = 0.f; v.y = 0.f; v.z = 0.f; v.w = 0.f; }
}

template <typename T>
__global__ void __launch_bounds__(kBlock)
k_rbgs(const T* __restrict__ vin, const T* __restrict__ rhs, T* __restrict__ vout,
       int N, long pitch, int row_lo, int row_hi, int R, int strips, int chunks,
       int row_parity, int bnd_lo, int bnd_hi)
{
    using V = typename VecOf<T>::type;
    constexpr int W = VecOf<T>::W;
    const Tile t = wave_tile(strips, chunks);
    if (!t.active) return;
    const Cols c = lane_cols<W>(t.strip, N, pitch);
    const int r0 = row_lo + t.chunk * R;
    const int r1 = min(r0 + R, row_hi);
    const T* pv = vin + c.col;
    const T* pb = rhs + c.col;
    T* po = vout + c.col;
    const V Z = vzero((V*)nullptr);
    const int cpar = (int)(c.col & 1);      // col parity of element 0 (W is even)

    // a row y may be read iff bnd_lo <= y <= bnd_hi; it is an unknown row iff
    // bnd_lo < y < bnd_hi.
    auto ldrow = [&](const T* base, int y) -> V {
        return vload<V>(base + (long)y * pitch, c.ld && y >= bnd_lo && y <= bnd_hi);
    };
    // red update of row y given old rows y-1, y, y+1 (identity on boundary rows)
    auto red = [&](const V& u, const V& m, const V& d, const V& b, int y) -> V {
        V o = gs_colour<0>(u, m, d, b, row_parity + y + cpar);
        mask_cols(o, c.col, N);
        return (y > bnd_lo && y < bnd_hi) ? o : Z;
    };

    // window: A = red(row r-1), B = red(row r), O1 = old(row r+1), O2 = old(row r+2)
    V o_m2 = ldrow(pv, r0 - 2), o_m1 = ldrow(pv, r0 - 1), o_0 = ldrow(pv, r0), o_p1 = ldrow(pv, r0 + 1);
    V A = red(o_m2, o_m1, o_0, ldrow(pb, r0 - 1), r0 - 1);
    V B = red(o_m1, o_0, o_p1, ldrow(pb, r0), r0);
    V bcur = ldrow(pb, r0);
    V O1 = o_p1;
    V O0 = o_0;   // old row r (needed as "up" when red-updating row r+1)
    for (int r = r0; r < r1; ++r) {
        const V O2 = ldrow(pv, r + 2);
        const V bnext = ldrow(pb, r + 1);
        const V Cn = red(O0, O1, O2, bnext, r + 1);          // red(row r+1)
        V o = gs_colour<1>(A, B, Cn, bcur, row_parity + r + cpar);   // black(row r)
        mask_cols(o, c.col, N);
        vstore<V>(po + (long)r * pitch, o, c.st);
        A = B; B = Cn; O0 = O1; O1 = O2; bcur = bnext;
    }
}

// =============================================================================
// K weighted-Jacobi sweeps in ONE pass over the data (temporal blocking):
//   vout = J^K(vin),  HBM traffic 3 sizeof(T) per point per K sweeps.
// The single-sweep kernel already moves its algorithmic bytes at ~71 % of the
// HBM peak; the only way past that is fewer bytes per update.  A wave marches
// down its chunk keeping, for every sweep level j < K, a 3-row window of level-j
// values in registers: loading input row y makes level-1 row y-1 computable,
// which makes level-2 row y-2 computable, ... and level-K row y-K is stored.
// Chunk edges recompute K halo rows per level (L2-served), strip edges use
// HL = ceil(K / W) halo lanes per side, so no wave depends on another wave.
// Each level performs exactly the arithmetic of one k_jacobi sweep, Dirichlet
// rows/columns are re-zeroed at every level: the result is bit-identical to K
// single sweeps.
// bnd_lo / bnd_hi: local indices of the global boundary rows (slabs); rows
// [max(row_lo-K, bnd_lo), min(row_hi+K-1, bnd_hi)] must exist.
// =============================================================================
// One level of a fused pass.  SM = 0: a weighted-Jacobi sweep.  SM = 1: half a
// red-black Gauss-Seidel sweep - odd levels update the red points (row + col
// even), even levels the black ones, so s sweeps are 2 s levels; `par` is the
// parity of (global row + first column of the vector).
template <typename T, int SM, int AR = 0>
__device__ __forceinline__ typename VecOf<T>::type
level_op(int j, const typename VecOf<T>::type& up, const typename VecOf<T>::type& cur,
         const typename VecOf<T>::type& dn, const typename VecOf<T>::type& bb, T c0, T c1, int par)
{
    if constexpr (SM == 0) {
        return jacobi_vec<T, AR>(up, cur, dn, bb, c0, c1);
    } else {
        return ((j - 1) & 1) ? gs_colour<1>(up, cur, dn, bb, par) : gs_colour<0>(up, cur, dn, bb, par);
    }
}

// EDGE = false: the wave's whole dependency cone lies strictly inside the grid
// (no Dirichlet row or column, no padding): every predicate, mask and select of
// the general body is statically true/absent.  Most waves take this path.
//
// One row step at rotation phase P.  Each level keeps its 3-row window in three
// fixed register slots; the incoming row overwrites the slot of the row that
// just died, so (oldest, middle, newest) = slots ((P+1)%3, (P+2)%3, P) and no
// window is ever shifted by register moves (the loop below runs P = 0,1,2).
// loads of one step: input row y and rhs row y-1
template <typename T, int K, bool EDGE, bool ZIN = false>
__device__ __forceinline__ void
fused_loads(typename VecOf<T>::type& in, typename VecOf<T>::type& bn, int y,
            const T* __restrict__ pv, const T* __restrict__ pb, long pitch, int r0, int r1, bool ld,
            int bnd_lo, int bnd_hi, int rd_lo, int rd_hi, bool zero_in, const FastOut& fo)
{
    using V = typename VecOf<T>::type;
    if constexpr (!EDGE) {
        // interior body: no predicate, no branch (see "branch-free stores" above)
#if MGX_SOFF_LOADS
        const unsigned srow = (unsigned)(y - fo.rb) * fo.pitch_bytes;
        if constexpr (ZIN) in = vzero((V*)nullptr);
        else in = bload_s((V*)nullptr, fo.in, fo.lane_off, srow);
        bn = bload_s((V*)nullptr, fo.rhs, fo.lane_off, srow - fo.pitch_bytes);
#else
        if constexpr (ZIN) in = vzero((V*)nullptr);
        else in = *reinterpret_cast<const V*>(pv + (long)y * pitch);
        bn = *reinterpret_cast<const V*>(pb + (long)(y - 1) * pitch);
#endif
    } else {
        // edge body: no branch either (see "branch-free loads" above).
        // [rd_lo, rd_hi]: rows inside the allocation and not beyond a boundary row;
        // zero_in: the input is known to be all zero (PS:613 coarse guess): do not read it
        const bool in_ok = !zero_in && y >= rd_lo && y <= rd_hi && y < r1 + K;
        // rhs rows are needed only where some level is: [r0-K+1, r1+K-1)
        const bool b_ok = (y - 1) > bnd_lo && (y - 1) < bnd_hi && (y - 1) >= rd_lo && (y - 1) <= rd_hi &&
                          y >= r0 - K + 2 && y < r1 + K;
        in = bload((V*)nullptr, fo.in, (ld && in_ok) ? (unsigned)(y - fo.rb) * fo.pitch_bytes + fo.lane_off : kOobOffset);
        bn = bload((V*)nullptr, fo.rhs, (ld && b_ok) ? (unsigned)(y - 1 - fo.rb) * fo.pitch_bytes + fo.lane_off : kOobOffset);
    }
}

// Steps per loop trip of the interior (branch-free) bodies.  s_waitcnt insertion is exact inside a
// straight-line trip and conservative at every loop back-edge, so longer trips keep more of the
// prefetched rows really in flight: 6 (two window rotations) for the shallow kernels, kBRing = 12
// for the deep ones.  The step count is rounded up to whole trips; the launch wrappers choose the
// chunk height so that at most one step is wasted (trip_rows below).
// double: 12 steps (at the loop head nothing is in flight, so a trip should be much longer than the
// prefetch distance); float: the packed-arithmetic bodies are register-bound and spill with long
// trips, they keep 3-step trips.
template <typename T> constexpr int trip_steps() { return 3; }
constexpr int kPfStages = 1;                 // prefetch slots per rotation phase (2 = six rows ahead: measured slower)
constexpr int kPrefetch = 3 * kPfStages;     // rows a marching wave loads ahead

template <typename T, int K, int SM, bool EDGE, int P, bool ZIN, int AR>
__device__ __forceinline__ void
fused_step(typename VecOf<T>::type (&lev)[K][3], typename VecOf<T>::type (&bw)[K],
           typename VecOf<T>::type (&nin)[kPfStages], typename VecOf<T>::type (&nbn)[kPfStages], int y,
           const T* __restrict__ pv, const T* __restrict__ pb, T* __restrict__ po, long pitch, long col, int N,
           int r0, int r1, bool ld, bool st, T c0, T c1, int bnd_lo, int bnd_hi, int rd_lo, int rd_hi, int par_c, bool zero_in,
           const FastOut& fo)
{
    using V = typename VecOf<T>::type;
    constexpr int S_OLD = (P + 1) % 3, S_MID = (P + 2) % 3, S_NEW = P;
    // level 0: input row y and rhs row y-1 were loaded three steps ago into this phase's
    // prefetch slot (each rotation phase owns one slot, so the prefetch queue needs no
    // register moves either); refill the slot with the rows of step y+3 before computing.
    // Three rows in flight per wave instead of one: a marching wave has no other way to
    // cover the HBM latency.
    if constexpr (EDGE) __builtin_amdgcn_sched_barrier(0);     // see cycle_step
    const V in = nin[0], bn = nbn[0];
#pragma unroll
    for (int q = 0; q + 1 < kPfStages; ++q) { nin[q] = nin[q + 1]; nbn[q] = nbn[q + 1]; }
    fused_loads<T, K, EDGE, ZIN>(nin[kPfStages - 1], nbn[kPfStages - 1], y + kPrefetch, pv, pb, pitch, r0, r1, ld, bnd_lo, bnd_hi, rd_lo, rd_hi, zero_in, fo);
    const ColMask cm = col_mask(col, N);
#pragma unroll
    for (int j = K - 1; j > 0; --j) bw[j] = bw[j - 1];
    if constexpr (SM == 0) bw[0] = vscale(c1, bn);       // Jacobi: the window holds c1 * b (see jacobi_vec_pre)
    else bw[0] = bn;
    lev[0][S_NEW] = in;
#pragma unroll
    for (int j = 1; j <= K; ++j) {
        // level-j row (y - j) from level-(j-1) rows (y-j-1, y-j, y-j+1) and rhs row y-j
        const int row = y - j;
        V o;
        if constexpr (SM == 0) o = jacobi_vec_pre<AR>(lev[j - 1][S_OLD], lev[j - 1][S_MID], lev[j - 1][S_NEW], bw[j - 1], c0, c1);
        else o = level_op<T, SM>(j, lev[j - 1][S_OLD], lev[j - 1][S_MID], lev[j - 1][S_NEW], bw[j - 1], c0, c1, par_c + row);
        if constexpr (EDGE) {                                                      // Dirichlet rows and columns stay zero
            const int rw = opaque_s(row);
            mask_sel(o, cm, !(rw > bnd_lo && rw < bnd_hi));
        }
        if (j < K) {
            lev[j][S_NEW] = o;
        } else {
            const unsigned at = (unsigned)(row - fo.rb) * fo.pitch_bytes + fo.lane_off;
            bstore(o, fo.out, (st && row >= r0 && row < r1) ? at : kOobOffset);
        }
    }
}

template <typename T, int K, int SM, bool EDGE, bool ZIN, int AR>
__device__ __forceinline__ void
fused_body(const T* __restrict__ pv, const T* __restrict__ pb, T* __restrict__ po, long pitch, long col, int N,
           int r0, int r1, bool ld, bool st, T c0, T c1, int bnd_lo, int bnd_hi, int rd_lo, int rd_hi, int row_parity, bool zero_in,
           const FastOut& fo)
{
    using V = typename VecOf<T>::type;
    const V Z = vzero((V*)nullptr);
    V lev[K][3];          // lev[j] = the three most recent level-j rows, rotating slots
    V bw[K];              // bw[j]  = rhs row y-1-j
#pragma unroll
    for (int j = 0; j < K; ++j) { lev[j][0] = Z; lev[j][1] = Z; lev[j][2] = Z; bw[j] = Z; }
    const int par_c = row_parity + (int)(col & 1);
    // steps y = r0-K .. r1+K-1, rounded up to a multiple of 3 (the extra steps
    // store nothing; in the EDGE body their loads are predicated, in the
    // interior body the caller guarantees five more rows exist below the cone:
    // two for the rounding, three for the prefetch)
    const int y0 = r0 - K;
    constexpr int kRound = trip_steps<T>();
    const int steps = (r1 + K - y0 + kRound - 1) / kRound * kRound;
    V nin[3][kPfStages], nbn[3][kPfStages];        // [rotation phase][queue position]
#pragma unroll
    for (int q = 0; q < kPrefetch; ++q)
        fused_loads<T, K, EDGE, ZIN>(nin[q % 3][q / 3], nbn[q % 3][q / 3], y0 + q, pv, pb, pitch, r0, r1, ld, bnd_lo, bnd_hi, rd_lo, rd_hi, zero_in, fo);
#define MGX_FSTEP(P, Y) fused_step<T, K, SM, EDGE, P, ZIN, AR>(lev, bw, nin[P], nbn[P], Y, pv, pb, po, pitch, col, N, r0, r1, ld, st, c0, c1, bnd_lo, bnd_hi, rd_lo, rd_hi, par_c, zero_in, fo)
    if constexpr (trip_steps<T>() == 12) {
        for (int y = y0; y < y0 + steps; y += 12) {
            MGX_FSTEP(0, y); MGX_FSTEP(1, y + 1); MGX_FSTEP(2, y + 2);
            MGX_FSTEP(0, y + 3); MGX_FSTEP(1, y + 4); MGX_FSTEP(2, y + 5);
            MGX_FSTEP(0, y + 6); MGX_FSTEP(1, y + 7); MGX_FSTEP(2, y + 8);
            MGX_FSTEP(0, y + 9); MGX_FSTEP(1, y + 10); MGX_FSTEP(2, y + 11);
        }
    } else {
        for (int y = y0; y < y0 + steps; y += 3) {
            MGX_FSTEP(0, y); MGX_FSTEP(1, y + 1); MGX_FSTEP(2, y + 2);
        }
    }
#undef MGX_FSTEP
}

template <typename T, int K, int SM = 0, int AR = 0>
__global__ void __launch_bounds__(kBlock)
k_jacobi_fused(const T* __restrict__ vin, const T* __restrict__ rhs, T* __restrict__ vout,
               int N, long pitch, int row_lo, int row_hi, int R, int strips, int chunks,
               T c0, T c1, int bnd_lo, int bnd_hi, int row_parity, int rows_alloc, int zero_in)
{
    constexpr int W = VecOf<T>::W;
    constexpr int HL = (K + W - 1) / W;           // halo lanes per side
    constexpr int OUT = kWave - 2 * HL;           // storing lanes per wave
    const Tile t = wave_tile(strips, chunks);
    if (!t.active) return;
    const int lane = threadIdx.x & 63;
    const int vx = t.strip * OUT - HL + lane;
    const long col = (long)vx * W;
    const bool ld = (vx >= 0) && (col + W <= pitch);
    const bool st = (lane >= HL) && (lane < kWave - HL) && (vx < N / W);
    const int r0 = row_lo + t.chunk * R;
    const int r1 = min(r0 + R, row_hi);
    const T* pv = vin + col;
    const T* pb = rhs + col;
    T* po = vout + col;
    // Rows that may be dereferenced at all: inside the allocation AND not beyond a
    // global boundary row (on a slab the allocation ends long before the boundary).
    const int rd_lo = max(bnd_lo, 0), rd_hi = min(bnd_hi, rows_alloc - 1);
    // wave-uniform: does everything the unpredicated body touches - rows r0-K-1 .. r1+K+4
    // (the step count is rounded up to a multiple of 3, plus three prefetched rows), vectors one
    // beyond the first and last lane - lie strictly inside the unknowns and inside the allocation?
    const int vx0 = t.strip * OUT - HL;
    const bool interior = (vx0 >= 1) && ((long)(vx0 + kWave) * W < N) &&
                          (r0 - K - 1 > bnd_lo) && (r0 - K - 1 >= 0) &&
                          (r1 + K + trip_steps<T>() + kPrefetch < bnd_hi) && (r1 + K + trip_steps<T>() + kPrefetch <= rows_alloc - 1);
    // both bodies store (and the edge body loads) through buffer descriptors that start at the first
    // row this wave can touch, with 32-bit offsets relative to it
    FastOut fo;
    fo.rb = max(r0 - K - 1, rd_lo);
    fo.crb = 0;
    fo.pitch_bytes = (unsigned)(pitch * (long)sizeof(T));
    fo.cpitch_bytes = 0;
    const unsigned f_bytes = rsrc_bytes(fo.rb, rows_alloc - 1, fo.pitch_bytes);
    const long f_at = (long)fo.rb * pitch;
    fo.out = make_rsrc(vout + f_at, f_bytes);
    fo.in = make_rsrc(vin + f_at, f_bytes);
    fo.rhs = make_rsrc(rhs + f_at, f_bytes);
    fo.cb = fo.out; fo.cz = fo.out; fo.ce = fo.out;   // unused here
    fo.lane_off = (unsigned)(col * (long)sizeof(T));
    fo.clane_off = 0;
    if (interior) {
        if (zero_in) fused_body<T, K, SM, false, true, AR>(pv, pb, po, pitch, col, N, r0, r1, true, st, c0, c1, bnd_lo, bnd_hi, rd_lo, rd_hi, row_parity, true, fo);
        else fused_body<T, K, SM, false, false, AR>(pv, pb, po, pitch, col, N, r0, r1, true, st, c0, c1, bnd_lo, bnd_hi, rd_lo, rd_hi, row_parity, false, fo);
    } else {
        fused_body<T, K, SM, true, false, AR>(pv, pb, po, pitch, col, N, r0, r1, ld, st, c0, c1, bnd_lo, bnd_hi, rd_lo, rd_hi, row_parity, zero_in != 0, fo);
    }
}

template <int K, int W> constexpr int fused_out_lanes() { return kWave - 2 * ((K + W - 1) / W); }

// =============================================================================
// residual r = b - A v   (PS:604-607: Av = LU v + D v = -(N+W+E+S) + 4 v)
// =============================================================================
__device__ __forceinline__ double2 residual_vec(const double2& up, const double2& cur, const double2& dn, const double2& bb)
{
    const double l = from_left(cur.y), r = from_right(cur.x);
    double2 o;
    o.x = bb.x - (-nbr(up.x, l, cur.y, dn.x) + 4.0 * cur.x);
    o.y = bb.y - (-nbr(up.y, cur.x, r, dn.y) + 4.0 * cur.y);
    return o;
}
__device__ __forceinline__ float4 residual_vec(const float4& up, const float4& cur, const float4& dn, const float4& bb)
{
    const NbrPairs t = nbr_pairs(up, cur, dn);
    const f32x2 p0 = {cur.x, cur.y}, p1 = {cur.z, cur.w}, b0 = {bb.x, bb.y}, b1 = {bb.z, bb.w};
    const f32x2 o0 = b0 - (-t.t0 + 4.f * p0);
    const f32x2 o1 = b1 - (-t.t1 + 4.f * p1);
    return make_float4(o0.x, o0.y, o1.x, o1.y);
}

// MODE 0: store r (same type).  MODE 1: accumulate sum r^2 only.
// MODE 2 (T = double): store (float)(r * inv_scale) into a float grid of pitch
// pitch_out AND accumulate sum r^2  (the mixed-precision defect, config 5).
template <typename T, int MODE>
__global__ void __launch_bounds__(kBlock)
k_residual(const T* __restrict__ vin, const T* __restrict__ rhs, void* __restrict__ out, long pitch_out,
           double* __restrict__ partial, double inv_scale,
           int N, long pitch, int row_lo, int row_hi, int R, int strips, int chunks, int rows_alloc)
{
    using V = typename VecOf<T>::type;
    constexpr int W = VecOf<T>::W;
    __shared__ double wsum[kWavesPerBlock];
    const Tile t = wave_tile(strips, chunks);
    double acc = 0.0;
    if (t.active) {
        const Cols c = lane_cols<W>(t.strip, N, pitch);
        const int r0 = row_lo + t.chunk * R;
        const int r1 = min(r0 + R, row_hi);
        const T* pv = vin + c.col;
        const T* pb = rhs + c.col;
        auto has = [&](int y) { return c.ld && y >= 0 && y < rows_alloc; };      // wave-uniform row test
        V up = vload<V>(pv + (long)(r0 - 1) * pitch, has(r0 - 1));
        V cur = vload<V>(pv + (long)r0 * pitch, has(r0));
        for (int r = r0; r < r1; ++r) {
            const V dn = vload<V>(pv + (long)(r + 1) * pitch, has(r + 1));
            const V bb = vload<V>(pb + (long)r * pitch, has(r));
            V o = residual_vec(up, cur, dn, bb);
            mask_cols(o, c.col, N);
            if (MODE == 0) {
                vstore<V>(reinterpret_cast<T*>(out) + c.col + (long)r * pitch, o, c.st);
            } else {
                if (c.st) {
                    if constexpr (W == 2) acc += (double)o.x * (double)o.x + (double)o.y * (double)o.y;
                    else acc += ((double)o.x * (double)o.x + (double)o.y * (double)o.y) +
                                ((double)o.z * (double)o.z + (double)o.w * (double)o.w);
                }
                if constexpr (MODE == 2 && W == 2) {
                    float2 f = make_float2((float)(o.x * inv_scale), (float)(o.y * inv_scale));
                    if (c.st) *reinterpret_cast<float2*>(reinterpret_cast<float*>(out) + c.col + (long)r * pitch_out) = f;
                }
            }
            up = cur; cur = dn;
        }
    }
    if (MODE != 0) {
        // wave shuffle reduction -> LDS -> one partial per block (deterministic:
        // no atomics, the final sum is a fixed-order second kernel)
        for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, kWave);
        if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = acc;
        __syncthreads();
        if (threadIdx.x == 0) {
            double s = 0.0;
            for (int w = 0; w < kWavesPerBlock; ++w) s += wsum[w];
            partial[blockIdx.x] = s;
        }
    }
}

// fixed-order final reduction of the per-block partials: out[0] = sum.  One workgroup of
// 1024 threads, four independent loads per thread in flight: a V(2,1) cycle at 8192^2 hands
// over 17 664 partials, which took 25 us with 256 threads and one load at a time.
constexpr int kReduceThreads = 1024;
static __global__ void __launch_bounds__(kReduceThreads) k_reduce_partials(const double* __restrict__ partial, int n, double* __restrict__ out)
{
    __shared__ double wsum[kReduceThreads / kWave];
    double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
    int i = threadIdx.x;
    for (; i + 3 * kReduceThreads < n; i += 4 * kReduceThreads) {
        a0 += partial[i];
        a1 += partial[i + kReduceThreads];
        a2 += partial[i + 2 * kReduceThreads];
        a3 += partial[i + 3 * kReduceThreads];
    }
    for (; i < n; i += kReduceThreads) a0 += partial[i];
    double acc = (a0 + a1) + (a2 + a3);
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, kWave);
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        double s = 0.0;
        for (int w = 0; w < kReduceThreads / kWave; ++w) s += wsum[w];
        out[0] = s;
    }
}

// =============================================================================
// restriction (PS:531-546 index pattern):  coarse(I,J) = w * (corners + 2 edges
// + 4 centre) centred on fine (2I, 2J), in the reference's summation order.
// FUSED: the fine field is the residual of (v, b), computed on the fly and never
// written (reads v + b, writes the coarse RHS: 2 + 1/4 values per fine point).
// !FUSED: the fine field is `rhs` itself (FMG right-hand sides, PS:641).
// ZERO_GUESS: also zero-fill the coarse solution array (PS:613) while here.
// A lane owns fine columns [col, col+W): W/2 coarse columns col/2 ...
// Coarse rows [crow_lo, crow_hi) are produced; fine local row of coarse local
// row I is 2*I + fine_row_off (0 on a single GPU; slab offset otherwise).
// =============================================================================
template <typename T> struct Trip { T l, c, r; };   // residual left / centre / right of a coarse column

template <typename T, bool FUSED>
__global__ void __launch_bounds__(kBlock)
k_restrict(const T* __restrict__ vin, const T* __restrict__ rhs, T* __restrict__ coarse, T* __restrict__ coarse_zero,
           int N, long pitch, long cpitch, int crow_lo, int crow_hi, int fine_row_off,
           int R, int strips, int chunks, T wgt)
{
    using V = typename VecOf<T>::type;
    constexpr int W = VecOf<T>::W;
    constexpr int CW = W / 2;               // coarse columns per lane
    const Tile t = wave_tile(strips, chunks);
    if (!t.active) return;
    const Cols c = lane_cols<W>(t.strip, N, pitch);
    const int I0 = crow_lo + t.chunk * R;
    const int I1 = min(I0 + R, crow_hi);
    const T* pv = vin + c.col;
    const T* pb = rhs + c.col;
    const long ccol = c.col / 2;
    const int NC = N / 2;

    // field row y (residual or plain), masked to the unknown columns
    auto row_plain = [&](int y) -> V { V o = vload<V>(pb + (long)y * pitch, c.ld); mask_cols(o, c.col, N); return o; };

    V up, cur, dn;          // v rows y-1, y, y+1 (FUSED only)
    int yf = 2 * I0 + fine_row_off - 1;     // first field row needed: 2*I0 - 1
    if (FUSED) {
        up = vload<V>(pv + (long)(yf - 1) * pitch, c.ld);
        cur = vload<V>(pv + (long)yf * pitch, c.ld);
    }
    auto next_field = [&](int y) -> V {
        if (FUSED) {
            dn = vload<V>(pv + (long)(y + 1) * pitch, c.ld);
            const V bb = vload<V>(pb + (long)y * pitch, c.ld);
            V o = residual_vec(up, cur, dn, bb);
            mask_cols(o, c.col, N);
            up = cur; cur = dn;
            return o;
        } else {
            return row_plain(y);
        }
    };
    // horizontal neighbours of the coarse columns this lane owns
    auto trips = [&](const V& f, Trip<T>* tr) {
        const T l = from_left(last(f));
        if constexpr (W == 2) { tr[0] = {l, f.x, f.y}; }
        else { tr[0] = {l, f.x, f.y}; tr[1] = {f.y, f.z, f.w}; }
    };

    Trip<T> top[CW], mid[CW], bot[CW];
    { const V f = next_field(yf); trips(f, top); }
    for (int I = I0; I < I1; ++I) {
        const int y = 2 * I + fine_row_off;
        { const V f = next_field(y); trips(f, mid); }
        { const V f = next_field(y + 1); trips(f, bot); }
        T o[CW];
#pragma unroll
        for (int k = 0; k < CW; ++k) {
            // PS:539-542 order: ((nw+ne)+sw)+se + 2*(((w+e)+n)+s) + 4*c
            T corners = top[k].l + top[k].r; corners = corners + bot[k].l; corners = corners + bot[k].r;
            T edges = mid[k].l + mid[k].r; edges = edges + top[k].c; edges = edges + bot[k].c;
            o[k] = wgt * ((corners + (T)2 * edges) + (T)4 * mid[k].c);
            if (ccol + k == 0 || ccol + k >= NC) o[k] = (T)0;
        }
        if (c.st) {
            T* pc = coarse + (long)I * cpitch + ccol;
            if constexpr (CW == 1) { pc[0] = o[0]; }
            else { *reinterpret_cast<float2*>(pc) = make_float2((float)o[0], (float)o[1]); }
            if (coarse_zero) {
                T* pz = coarse_zero + (long)I * cpitch + ccol;
                if constexpr (CW == 1) { pz[0] = (T)0; }
                else { *reinterpret_cast<float2*>(pz) = make_float2(0.f, 0.f); }
            }
        }
#pragma unroll
        for (int k = 0; k < CW; ++k) top[k] = bot[k];
    }
}

// =============================================================================
// bilinear prolongation (PS:337-425) with the coarse-grid correction add
// (PS:620-624) fused:  ADD: v += P e  (in place);  !ADD: v = P e  (FMG, PS:645).
// A lane owns fine columns [col, col+W) of fine rows [row_lo,row_hi); coarse
// values e(I, col/2 .. col/2 + W/2) come straight from the coarse array, whose
// own zero ring supplies the reference's boundary special cases (PS:341-390).
// Fine local row y maps to coarse local row (y - fine_row_off)/2.
// =============================================================================
template <typename T, bool ADD>
__global__ void __launch_bounds__(kBlock)
k_prolong(T* __restrict__ v, const T* __restrict__ coarse, int N, long pitch, long cpitch,
          int row_lo, int row_hi, int fine_row_off, int R, int strips, int chunks)
{
    using V = typename VecOf<T>::type;
    constexpr int W = VecOf<T>::W;
    constexpr int CW = W / 2;
    const Tile t = wave_tile(strips, chunks);
    if (!t.active) return;
    const Cols c = lane_cols<W>(t.strip, N, pitch);
    const int r0 = row_lo + t.chunk * R;
    const int r1 = min(r0 + R, row_hi);
    T* pv = v + c.col;
    const long ccol = c.col / 2;
    const bool cld = c.ld && (ccol + CW < cpitch);

    // coarse row I: values at coarse columns ccol .. ccol+CW (CW+1 of them)
    auto crow = [&](int I, T* e) {
        const T* p = coarse + (long)I * cpitch + ccol;
#pragma unroll
        for (int k = 0; k <= CW; ++k) e[k] = cld ? p[k] : (T)0;
    };
    for (int r = r0; r < r1; ++r) {
        const int y = r - fine_row_off;
        const int I = y >> 1;
        T a[CW + 1], b2[CW + 1];
        crow(I, a);
        V add;
        T o[W];
        if ((y & 1) == 0) {             // PS:398-402, 410-414
#pragma unroll
            for (int k = 0; k < CW; ++k) { o[2 * k] = a[k]; o[2 * k + 1] = (T)0.5 * (a[k] + a[k + 1]); }
        } else {                        // PS:404-408, 416-420: ((NW + SW) + NE) + SE
            crow(I + 1, b2);
#pragma unroll
            for (int k = 0; k < CW; ++k) {
                o[2 * k] = (T)0.5 * (a[k] + b2[k]);
                o[2 * k + 1] = (T)0.25 * (((a[k] + b2[k]) + a[k + 1]) + b2[k + 1]);
            }
        }
        if constexpr (W == 2) add = make_double2(o[0], o[1]);
        else add = make_float4(o[0], o[1], o[2], o[3]);
        mask_cols(add, c.col, N);
        if (ADD) {
            const V old = vload<V>(pv + (long)r * pitch, c.st);
            if constexpr (W == 2) { add.x = old.x + add.x; add.y = old.y + add.y; }
            else { add.x = old.x + add.x; add.y = old.y + add.y; add.z = old.z + add.z; add.w = old.w + add.w; }
        }
        vstore<V>(pv + (long)r * pitch, add, c.st);
    }
}

// =============================================================================
// k_jacobi_cycle: k_jacobi_fused with the V-cycle's transfer operators folded
// into the same pass over the finest data, so that v and b are not re-read for
// them (single-GPU hierarchy: coarse row I sits on fine row 2I of the same grid):
//   PRE  = 1 : the input row is  v + P e  (PS:620-624), e the coarse correction;
//              the corrected v is consumed on the fly and never stored.
//   POST = 1 : after the K-th sweep the residual b - A v (PS:604-607) of the new
//              iterate is formed in registers and restricted (PS:531-546) into
//              the coarse right-hand side; the coarse guess is zeroed (PS:613).
//   POST = 2 : after the K-th sweep sum (b - A v)^2 over the chunk -> partial[]
//              (the residual norm the solve loop reports after every cycle).
// K >= 1.  Every stage performs the arithmetic of the stand-alone kernel it
// replaces, in the same order: results are bit-identical (tests).
// Needs XC = K + {0,2,1}[POST] halo columns per side and one extra level window.
// =============================================================================
template <int K, int POST> constexpr int cycle_halo_cols() { return K + (POST == 1 ? 2 : (POST == 2 ? 1 : 0)); }
template <int K, int POST, int W> constexpr int cycle_out_lanes() { return kWave - 2 * ((cycle_halo_cols<K, POST>() + W - 1) / W); }

// per-wave state of a folded pass that is not a level window
// (plain scalars, assigned member by member: aggregate copies of a 24-byte
// Trip<double> were lowered to scratch memcpys)
template <typename T, int CW> struct CycleState {
    // POST 1: what is left of rows 2I-1 (top) and 2I (mid) of the coarse row being assembled, per
    // coarse column of the lane, as the partial sums of PS:539-542 in its order:
    //   ct = nw + ne, tc = n (top row);  em = ((w + e) + n), mc = c (mid row)
    // four values instead of the six residuals: the folded restriction is register-bound
    T ct[CW], tc[CW], em[CW], mc[CW];
    double acc;                 // POST 2: sum of r^2
};

// The rows that exist in memory, in GLOBAL row numbers (whole grid: 0..N and 0..N/2; a slab
// of a row-decomposed grid: its rows incl. halos - the kernel is then given base pointers
// moved back by row0 rows, so that global numbers index them), and the coarse rows POST = 1
// may write (whole grid: 1..N/2-1; slab: the coarse rows this rank owns).
struct CycleWin {
    int row_first, row_last;        // fine rows [row_first, row_last] exist
    int crow_first, crow_last;      // coarse rows [crow_first, crow_last] exist
    int emit_lo, emit_hi;           // coarse rows [emit_lo, emit_hi) are written
};

struct CycleArgs {              // what the stages besides the smoother need (uniform)
    long cpitch; int NC;
    int r0, r1, y_end;
    bool zero_in;               // the input iterate is all zero: do not read it
    CycleWin win;
};

// one row step of k_jacobi_cycle at window-rotation phase P (see fused_step)
// PRE: the coarse correction values one fine row needs: coarse row I = y>>1 (a) and, for
// odd y, row I+1 (b); CW+1 columns each
template <typename T, int CW> struct PreFetch { T a[CW + 1], b[CW + 1]; };

template <typename T, bool EDGE>
__device__ __forceinline__ void
coarse_loads(PreFetch<T, VecOf<T>::W / 2>& pe, int y, const T* __restrict__ coarse_e, long cpitch, long ccol, int N, bool cld,
             const CycleWin& win, const FastOut& fo)
{
    constexpr int CW = VecOf<T>::W / 2;
    const int I = y >> 1;
    if constexpr (!EDGE) {
        // (coarse rows beyond the wave's cone: the nearest one that exists, as the fine rows in cycle_loads)
        const int Ia = min(max(I, win.crow_first), win.crow_last), Ib = min(max(I + 1, win.crow_first), win.crow_last);
        const unsigned srow_a = (unsigned)(Ia - fo.crb) * fo.cpitch_bytes, srow_b = (unsigned)(Ib - fo.crb) * fo.cpitch_bytes;
#pragma unroll
        for (int k = 0; k <= CW; ++k) pe.a[k] = bload1_s((T*)nullptr, fo.ce, fo.clane_off + (unsigned)(k * sizeof(T)), srow_a);
        // the row below is only used by odd fine rows; loading it always keeps the step branch-free
#pragma unroll
        for (int k = 0; k <= CW; ++k) pe.b[k] = bload1_s((T*)nullptr, fo.ce, fo.clane_off + (unsigned)(k * sizeof(T)), srow_b);
    } else {
        const bool cl = cld && y > 0 && y < N && I >= win.crow_first && I + (y & 1) <= win.crow_last;
        const bool cl2 = cl && (y & 1);
        const unsigned at = (unsigned)(I - fo.crb) * fo.cpitch_bytes + fo.clane_off;
#pragma unroll
        for (int k = 0; k <= CW; ++k) pe.a[k] = bload1((T*)nullptr, fo.ce, cl ? at + (unsigned)(k * sizeof(T)) : kOobOffset);
#pragma unroll
        for (int k = 0; k <= CW; ++k) pe.b[k] = bload1((T*)nullptr, fo.ce, cl2 ? at + fo.cpitch_bytes + (unsigned)(k * sizeof(T)) : kOobOffset);
    }
}

// loads of one step of k_jacobi_cycle (whole grids: rows 0..N exist)
template <typename T, bool EDGE, bool ZIN = false>
__device__ __forceinline__ void
cycle_loads(typename VecOf<T>::type& in, typename VecOf<T>::type& bn, int y,
            const T* __restrict__ pv, const T* __restrict__ pb, long pitch, int N, int y_end, bool ld, bool zero_in,
            const CycleWin& win, const FastOut& fo)
{
    using V = typename VecOf<T>::type;
    if constexpr (!EDGE) {
        // interior body: no predicate, no branch (ZIN: the input is known to be all zero)
        // (rows beyond the wave's cone - see `interior` in k_jacobi_cycle - come from the nearest row that exists)
        const int yi = min(y, win.row_last), yb = max(min(y - 1, win.row_last), win.row_first);
        if constexpr (ZIN) in = vzero((V*)nullptr);
        else in = bload_s((V*)nullptr, fo.in, fo.lane_off, (unsigned)(yi - fo.rb) * fo.pitch_bytes);
        bn = bload_s((V*)nullptr, fo.rhs, fo.lane_off, (unsigned)(yb - fo.rb) * fo.pitch_bytes);
    } else {
        // edge body: no branch either - a row or lane that must not be read reads 0 through the
        // descriptor (win.row_first >= 0 and win.row_last <= N: the window also keeps y inside the grid)
        const bool in_ok = !zero_in && y >= win.row_first && y <= win.row_last && y < y_end;
        const bool b_ok = (y - 1) > 0 && (y - 1) < N && (y - 1) >= win.row_first && (y - 1) <= win.row_last && y <= y_end;
        in = bload((V*)nullptr, fo.in, (ld && in_ok) ? (unsigned)(y - fo.rb) * fo.pitch_bytes + fo.lane_off : kOobOffset);
        bn = bload((V*)nullptr, fo.rhs, (ld && b_ok) ? (unsigned)(y - 1 - fo.rb) * fo.pitch_bytes + fo.lane_off : kOobOffset);
    }
}

// ---- the rhs window of a deep folded pass in LDS -------------------------------------------------
// A pass of K levels uses every rhs row at K (+1 with a residual stage) consecutive steps: a delay
// line of K + 1 vectors per lane.  In registers that is 4 (K + 1) VGPRs; with the level windows
// (12 (K + 1)), the prefetch slots and the transfer state the 10-level folded passes needed more than
// 256 and ran at one wave per SIMD with accumulator-register spills.  For deep passes the delay line
// therefore lives in LDS: each wave owns a ring of kBRing rows of 1 KiB (64 lanes x 16 B), row y at
// slot y mod kBRing; the step loop is unrolled kBRing-fold so that every slot is a compile-time
// ds_read_b128 / ds_write_b128 offset from ONE per-lane address (no address arithmetic, no register
// moves); no barrier (a wave only ever touches its own ring).  LDS instructions do not occupy the
// vector ALU, which is what these passes are bound by.  Same values, same order: same bits.
constexpr int kBRing = 12;                     // >= K + 1 for K <= 10, multiple of the 3 rotation phases
// (float: from 10 levels - the 8-level float bodies fit their registers with the window in them and keep 3-step trips)
template <typename T, int K, int POST, int SM> constexpr bool cycle_b_in_lds() { return (sizeof(T) == 8 && K >= 8) || (sizeof(T) == 4 && K >= 10); }
// Two level chains per step (round 3; built, bit-identical, measured, OFF).  A row step of a K-level pass is a serial
// chain: level j needs level j-1's row of this very step, ten times over, and a wave's two points per lane are all the
// independent work there is; the counters (profiles/r03a_sq_summary.md: vector ALU busy 0.64-0.67 with 1.6 waves per
// SIMD, a third of the wave time stalled on issue) suggested more instruction-level parallelism.  With a SKEW the upper
// levels run ONE STEP BEHIND: levels SK+1 .. K of step y work on rows y-j-1 from what level SK left in its window at step
// y-1, so chains A = levels 1..SK and B = levels SK+1..K of one step are independent and their instructions interleave.
// Cost: one more row step per chunk, one more row of level SK alive, one more entry of the rhs delay line.  Same
// operations on the same values in the same order: same bits (188 parity tests).  Measured against the serial chain inside
// one GPU call: 8192^2 pass 0.412-0.418 against 0.386-0.413 ms, 16384^2 1.48 against 1.51, 4096^2 equal: the second wave
// of the SIMD already fills what a lone chain leaves, so it stays off (-DMGX_SKEW=1 builds it).
#ifndef MGX_SKEW
#define MGX_SKEW 0
#endif
#ifndef MGX_RING_AHEAD
#define MGX_RING_AHEAD 1     // levels a ring entry is fetched ahead of its use
#endif
#ifndef MGX_PRIO_ALT
#define MGX_PRIO_ALT 0      // experiment knob (round 3, within noise): the two waves of a SIMD alternate at the higher issue priority, trip by trip
#endif
template <typename T, int K, int PRE, int POST, int SM, int AR> constexpr int cycle_skew()
{
    // (not where the interleaved chains' registers do not fit: these two variants would spill 16-24 B)
    if (sizeof(T) == 8 && K == 10 && PRE == 1 && POST == 0 && SM == 0 && AR == 0) return 0;
    if (sizeof(T) == 4 && K == 10 && PRE == 0 && POST == 1 && SM == 0) return 0;
    return (MGX_SKEW != 0 && cycle_b_in_lds<T, K, POST, SM>()) ? K / 2 : 0;
}
// The rhs window holds c1 * b (one multiplication per point and step instead of one per point, step and LEVEL: 18 of the
// ~200 vector instructions of a 10-level row step); the residual stage of POST needs b itself, K steps after it was
// loaded: a second delay line, kRawLds steps deep in LDS (a ring of its own: 24 KiB per workgroup, two workgroups still
// fit a CU's 160 KiB) and K - kRawLds steps in registers.  Same values into the same operations: same bits.
constexpr int kRawLds = 6;                     // divides kBRing: every slot a compile-time offset
#ifndef MGX_RAWQ
#define MGX_RAWQ 1
#endif
template <typename T, int K, int PRE, int POST, int SM, int AR> constexpr bool cycle_rawq();
template <typename T> struct LdsVec;
template <> struct LdsVec<double> { typedef double v __attribute__((ext_vector_type(2))); };
template <> struct LdsVec<float> { typedef float v __attribute__((ext_vector_type(4))); };
template <typename T> using lds_vec_ptr = __attribute__((address_space(3))) typename LdsVec<T>::v*;
__device__ __forceinline__ void ring_put(lds_vec_ptr<double> r, int slot, const double2& v)
{
    LdsVec<double>::v t = {v.x, v.y};
    r[slot * kWave] = t;
}
__device__ __forceinline__ double2 ring_get(lds_vec_ptr<double> r, int slot)
{
    const LdsVec<double>::v t = r[slot * kWave];
    return make_double2(t.x, t.y);
}
__device__ __forceinline__ void ring_put(lds_vec_ptr<float> r, int slot, const float4& v)
{
    LdsVec<float>::v t = {v.x, v.y, v.z, v.w};
    r[slot * kWave] = t;
}
__device__ __forceinline__ float4 ring_get(lds_vec_ptr<float> r, int slot)
{
    const LdsVec<float>::v t = r[slot * kWave];
    return make_float4(t.x, t.y, t.z, t.w);
}
constexpr int ring_slot(int m) { return ((m % kBRing) + kBRing) % kBRing; }
template <typename T, int K, int PRE, int POST, int SM, int AR> constexpr bool cycle_rawq()
{
    if (MGX_RAWQ == 0) return false;
    if (PRE != 0 || sizeof(T) != 8) return false;     // (registers: <double,10,1,2> would spill 76 B with the delay line)
    if (POST == 2 && AR == 0 && K == 10) return false;  // (12 B)
    return cycle_b_in_lds<T, K, POST, SM>() && SM == 0 && POST != 0 && K > kRawLds && cycle_skew<T, K, PRE, POST, SM, AR>() == 0;
}
// rows a folded pass loads ahead of itself (one register slot pair per row in flight).  Three
// everywhere, except the deep pre-smoothing passes with the restriction stage: they are the most
// register-hungry kernels of the library, a step of theirs is ~800 vector instructions long, and two
// rows in flight keep them at two waves per SIMD without spills.
// (the deep edge bodies, with their masks on top, too: the 10-level correction pass spilled with three)
#ifndef MGX_PFD_DEEP
#define MGX_PFD_DEEP 0      // experiment knob: rows in flight in the interior bodies of the deep passes (0: 2 with the restriction stage, else 3)
#endif
template <bool BL, int POST, bool EDGE = false> constexpr int cycle_pfd()
{
    if (BL && !EDGE && MGX_PFD_DEEP > 0) return MGX_PFD_DEEP;
    return (BL && (POST == 1 || EDGE)) ? 2 : kPrefetch;
}
constexpr int kPrefetchMax = MGX_PFD_DEEP > kPrefetch ? MGX_PFD_DEEP : kPrefetch;
// rows the coarse correction (PRE) is fetched ahead.  vmcnt counts in issue order, so waiting for a
// coarse row fetched ONE step ago also waits for every fine row issued before it: with a one-step
// coarse prefetch the three-row fine prefetch was worth one row.  The interior bodies fetch the
// coarse rows as far ahead as the fine ones (8 VGPRs per extra row in double).
// (edge bodies: as deep as their fine rows too, except in the passes without a residual stage - the 10-level
// one is at 256 registers and would spill)
template <typename T, bool BL, bool EDGE, int POST> constexpr int cycle_cpfd() { return (BL && !(EDGE && POST == 0)) ? cycle_pfd<BL, POST, EDGE>() : 1; }

// RP: phase of the step inside the kBRing-fold unrolled loop (BL) or inside the 3-fold one (!BL);
// the window-rotation phase is RP % 3 either way
template <typename T, int K, int PRE, int POST, int SM, bool EDGE, int RP, bool BL, bool ZIN, int AR>
__device__ __forceinline__ void
cycle_step(typename VecOf<T>::type (&lev)[K + 1][3], typename VecOf<T>::type (&bw)[BL ? 1 : K + 1], lds_vec_ptr<T> ring,
           lds_vec_ptr<T> ring2, typename VecOf<T>::type (&rawq)[cycle_rawq<T, K, PRE, POST, SM, AR>() ? kBRing : 1],
           typename VecOf<T>::type (&nin)[kPfStages], typename VecOf<T>::type (&nbn)[kPfStages], PreFetch<T, VecOf<T>::W / 2>& pe,   // pe: this step's slot
           CycleState<T, VecOf<T>::W / 2>& cs, int y,
           const T* __restrict__ pv, const T* __restrict__ pb, T* __restrict__ po,
           const T* __restrict__ coarse_e, T* __restrict__ coarse_b, T* __restrict__ coarse_zero, T wgt,
           long pitch, long col, long ccol, int N, const CycleArgs& ca, bool ld, bool cld, bool st, T c0, T c1,
           const FastOut& fo)
{
    using V = typename VecOf<T>::type;
    constexpr int W = VecOf<T>::W;
    constexpr int CW = W / 2;
    constexpr int P = RP % 3;
    constexpr int S_OLD = (P + 1) % 3, S_MID = (P + 2) % 3, S_NEW = P;
    constexpr int bnd_lo = 0;
    const int bnd_hi = N;
    const int r0 = ca.r0, r1 = ca.r1;

    // (edge bodies: left alone, the scheduler computes the load offsets and row masks of all the steps
    // of a trip at its top - some 70 VGPRs and 150 spilled SGPRs - so each step is fenced)
    if constexpr (EDGE) __builtin_amdgcn_sched_barrier(0);
    // input row y and rhs row y-1 were loaded during the previous step (software
    // prefetch, see fused_step); issue the next step's loads before computing
    V in = nin[0];
    const V bn = nbn[0];
#pragma unroll
    for (int q = 0; q + 1 < kPfStages; ++q) { nin[q] = nin[q + 1]; nbn[q] = nbn[q + 1]; }
    cycle_loads<T, EDGE, ZIN>(nin[kPfStages - 1], nbn[kPfStages - 1], y + cycle_pfd<BL, POST, EDGE>(), pv, pb, pitch, N, ca.y_end, ld, ca.zero_in, ca.win, fo);
    const ColMask cm = col_mask(col, N);
    if (PRE) {
        // v + P e on unknown rows, exactly as k_prolong<T,true> (PS:620-624).  The coarse
        // values of row y were fetched during the previous step (pe.a = coarse row y>>1,
        // pe.b = the row below it); fetch the next row's before using these.
        T a[CW + 1], b2[CW + 1], o[W];
#pragma unroll
        for (int k = 0; k <= CW; ++k) { a[k] = pe.a[k]; b2[k] = pe.b[k]; }
        coarse_loads<T, EDGE>(pe, y + cycle_cpfd<T, BL, EDGE, POST>(), coarse_e, ca.cpitch, ccol, N, cld, ca.win, fo);
        {
            // branch-free: both row parities evaluated (same expressions, same order), one selected
            const bool even = (y & 1) == 0;
#pragma unroll
            for (int k = 0; k < CW; ++k) {
                const T ab = a[k] + b2[k];
                const T ev1 = (T)0.5 * (a[k] + a[k + 1]);
                const T od0 = (T)0.5 * ab;
                const T od1 = (T)0.25 * ((ab + a[k + 1]) + b2[k + 1]);
                o[2 * k] = even ? a[k] : od0;
                o[2 * k + 1] = even ? ev1 : od1;
            }
        }
        V add;
        if constexpr (W == 2) add = make_double2(o[0], o[1]);
        else add = make_float4(o[0], o[1], o[2], o[3]);
        if constexpr (EDGE) mask_sel(add, cm, false);
        if constexpr (W == 2) { in.x = in.x + add.x; in.y = in.y + add.y; }
        else { in.x = in.x + add.x; in.y = in.y + add.y; in.z = in.z + add.z; in.w = in.w + add.w; }
    }
    // Jacobi passes without a residual stage keep c1 * b in the window (see jacobi_vec_pre);
    // the residual of POST needs b itself
    constexpr bool RAWQ = cycle_rawq<T, K, PRE, POST, SM, AR>();
    constexpr bool PREMUL = (SM == 0 && POST == 0) || RAWQ;
    V b0;                                             // rhs row y-1 (times c1 when PREMUL): bw[0]
    if constexpr (PREMUL) b0 = vscale(c1, bn);
    else b0 = bn;
    if constexpr (BL) {
        // the compiler must not carry ring contents in registers from step to step (it would,
        // every offset being a constant): make the address opaque once per step
        asm volatile("" : "+v"(ring));
        ring_put(ring, ring_slot(RP), b0);
        if constexpr (RAWQ) {
            // b itself, for the residual stage K steps from now: kRawLds steps in the second ring (read the slot's old
            // content - rhs row y-1-kRawLds - before overwriting it: LDS operations of a wave complete in order), the rest in registers
            asm volatile("" : "+v"(ring2));
            rawq[ring_slot(RP)] = ring_get(ring2, RP % kRawLds);
            ring_put(ring2, RP % kRawLds, bn);
        }
    } else {
#pragma unroll
        for (int j = K; j > 0; --j) bw[j] = bw[j - 1];
        bw[0] = b0;
    }
    // bw[j] = rhs row y-1-j.  From the LDS ring (BL) the reads are issued kRingAhead levels before
    // their use and pinned there: left to itself the compiler sinks every ds_read_b128 to ~5
    // instructions in front of its consumer, and the ~100-cycle LDS latency sat exposed in every one
    // of the K levels of the serial level-to-level chain.
    constexpr int kRingAhead = MGX_RING_AHEAD;
    constexpr int SK = cycle_skew<T, K, PRE, POST, SM, AR>();      // 0: one serial chain of K levels; else levels SK+1..K run one step behind
    constexpr int SKD = SK > 0 ? 1 : 0;
    constexpr int MLAST = ((POST && !RAWQ) ? K : K - 1) + SKD;   // last window entry this step consumes
    V rq[BL ? K + 2 : 1];                              // BL: ring values in flight (a sliding window is live)
    auto bwin = [&](auto jc) -> V {
        constexpr int j = decltype(jc)::value;
        if constexpr (j == 0) return b0;
        else if constexpr (BL) return rq[j];
        else return bw[j];
    };
    // the row of level SK that dies in this step (its slot takes the new row): chain B's first level still needs it
    V dying = lev[SK][S_NEW];
    lev[0][S_NEW] = in;
    // level j of this step: row y-j (chain A / no skew) or y-j-1 (chain B) from level j-1's window and rhs row `row`
    auto level = [&](auto jc, auto bc) {
        constexpr int j = decltype(jc)::value;
        constexpr bool inB = decltype(bc)::value;
        const int row = y - j - (inB ? 1 : 0);
        const V cb = bwin(std::integral_constant<int, j - 1 + (inB ? 1 : 0)>{});
        // (oldest, middle, newest) rows of level j-1: chain B's first level reads them as they stood BEFORE this step
        const V& lo = (inB && j == SK + 1) ? dying : lev[j - 1][S_OLD];
        const V& mi = (inB && j == SK + 1) ? lev[j - 1][S_OLD] : lev[j - 1][S_MID];
        const V& hi = (inB && j == SK + 1) ? lev[j - 1][S_MID] : lev[j - 1][S_NEW];
        V o;
        if constexpr (PREMUL) o = jacobi_vec_pre<AR>(lo, mi, hi, cb, c0, c1);
        else o = level_op<T, SM, AR>(j, lo, mi, hi, cb, c0, c1, row + (int)(col & 1));
        if constexpr (EDGE) {                                                      // Dirichlet rows and columns stay zero
            const int rw = opaque_s(row);
            mask_sel(o, cm, !(rw > bnd_lo && rw < bnd_hi));
        }
        if constexpr (j == K) {
            const unsigned at = (unsigned)(row - fo.rb) * fo.pitch_bytes + fo.lane_off;
            bstore(o, fo.out, (st && row >= r0 && row < r1) ? at : kOobOffset);
        }
        if (j < K || POST) lev[j][S_NEW] = o;
    };
    if constexpr (SK == 0) {
        if constexpr (BL) {
            static_for<1, (kRingAhead < MLAST ? kRingAhead : MLAST) + 1>([&](auto mc) {
                constexpr int m = decltype(mc)::value;
                rq[m] = ring_get(ring, ring_slot(RP - m));
            });
        }
        static_for<1, K + 1>([&](auto jc) {
            constexpr int j = decltype(jc)::value;
            if constexpr (BL && j + kRingAhead <= MLAST) {
                rq[j + kRingAhead] = ring_get(ring, ring_slot(RP - (j + kRingAhead)));
                __builtin_amdgcn_sched_barrier(0);
            }
            level(jc, std::false_type{});
        });
    } else {
        // pairs (A_i, B_i) = (level i, level SK + i): independent of each other inside a pair, so the scheduler
        // interleaves them; the ring entries of the next pair (A: row y-1-i, B: row y-1-(SK+i+1)) are fetched a pair ahead
        constexpr int NB = K - SK;
        constexpr int NP = SK > NB ? SK : NB;
        rq[SK + 1] = ring_get(ring, ring_slot(RP - (SK + 1)));                 // B_1's rhs row (A_1 uses b0)
        static_for<1, NP + 1>([&](auto ic) {
            constexpr int i = decltype(ic)::value;
            if constexpr (i + 1 <= SK) rq[i] = ring_get(ring, ring_slot(RP - i));                          // A_(i+1): bwin(i)
            if constexpr (i + 1 <= NB) rq[SK + i + 1] = ring_get(ring, ring_slot(RP - (SK + i + 1)));    // B_(i+1)
            if constexpr (POST != 0 && i == NP) rq[K + 1] = ring_get(ring, ring_slot(RP - (K + 1)));       // the residual stage's row
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (i <= SK) level(std::integral_constant<int, i>{}, std::false_type{});
            if constexpr (i <= NB) level(std::integral_constant<int, SK + i>{}, std::true_type{});
        });
    }
    if (POST) {
        // residual of the new iterate on row rho = y-K-1 (rows rho-1, rho, rho+1 of level K; one step later with the skew)
        const int rho = y - K - 1 - SKD;
        V rawb;                                       // rhs row rho
        if constexpr (RAWQ) rawb = rawq[ring_slot(RP - (K - kRawLds))];
        else rawb = bwin(std::integral_constant<int, K + SKD>{});
        V res = residual_vec(lev[K][S_OLD], lev[K][S_MID], lev[K][S_NEW], rawb);
        if constexpr (EDGE) {
            const int rw = opaque_s(rho);
            mask_sel(res, cm, !(rw > bnd_lo && rw < bnd_hi));
        }
        if (POST == 2) {
            double r2;
            if constexpr (W == 2) r2 = (double)res.x * (double)res.x + (double)res.y * (double)res.y;
            else r2 = ((double)res.x * (double)res.x + (double)res.y * (double)res.y) +
                      ((double)res.z * (double)res.z + (double)res.w * (double)res.w);
            cs.acc += (st && rho >= r0 && rho < r1) ? r2 : 0.0;          // + 0.0 is exact: same sum, no branch
        } else {
            T cl[CW], cc[CW], cr[CW];          // this row's residual left / centre / right per coarse column
            const T l = from_left(last(res));
            if constexpr (W == 2) { cl[0] = l; cc[0] = res.x; cr[0] = res.y; }
            else { cl[0] = l; cc[0] = res.x; cr[0] = res.y; cl[1] = res.y; cc[1] = res.z; cr[1] = res.w; }
            // rho odd (= 2I+1) closes coarse row I (top = 2I-1, mid = 2I, bot = this row) and
            // becomes the next top; rho even becomes mid.  Written with selects, not
            // branches: a branch on the run-time parity made the compiler index the
            // state as a stack array (scratch).
            const bool odd = (rho & 1) != 0;
            const int I = (rho - 1) >> 1;
            const bool emit = odd && (2 * I >= r0) && (2 * I < r1) && I >= ca.win.emit_lo && I < ca.win.emit_hi;
            T o[CW];
#pragma unroll
            for (int k = 0; k < CW; ++k) {
                // PS:539-542 order: ((nw+ne)+sw)+se + 2*(((w+e)+n)+s) + 4*c
                T corners = cs.ct[k] + cl[k]; corners = corners + cr[k];
                const T edges = cs.em[k] + cc[k];
                o[k] = wgt * ((corners + (T)2 * edges) + (T)4 * cs.mc[k]);
                if constexpr (EDGE) o[k] = (ccol + k == 0 || ccol + k >= ca.NC) ? (T)0 : o[k];
            }
            {
                const unsigned at = (st && emit) ? ((unsigned)(I - fo.crb) * fo.cpitch_bytes + fo.clane_off) : kOobOffset;
                if constexpr (CW == 1) { bstore8(o[0], fo.cb, at); bstore8((T)0, fo.cz, at); }
                else { bstore8(make_float2((float)o[0], (float)o[1]), fo.cb, at); bstore8(make_float2(0.f, 0.f), fo.cz, at); }
            }
#pragma unroll
            for (int k = 0; k < CW; ++k) {
                const T lr = cl[k] + cr[k];                    // nw + ne of the next coarse row / w + e of this one
                const T em = lr + cs.tc[k];                    // (w + e) + n   (uses the OLD top centre)
                cs.em[k] = odd ? cs.em[k] : em;
                cs.mc[k] = odd ? cs.mc[k] : cc[k];
                cs.ct[k] = odd ? lr : cs.ct[k];
                cs.tc[k] = odd ? cc[k] : cs.tc[k];
            }
        }
    }
}

template <typename T, int K, int PRE, int POST, int SM, bool EDGE, bool ZIN, int AR>
__device__ __forceinline__ double
cycle_body(const T* __restrict__ pv, const T* __restrict__ pb, T* __restrict__ po,
           const T* __restrict__ coarse_e, T* __restrict__ coarse_b, T* __restrict__ coarse_zero, T wgt,
           long pitch, long cpitch, long col, int N, int r0, int r1, bool ld, bool st, T c0, T c1, bool zero_in,
           const CycleWin& win, lds_vec_ptr<T> ring, lds_vec_ptr<T> ring2, const FastOut& fo)
{
    constexpr bool BL = cycle_b_in_lds<T, K, POST, SM>();
    constexpr bool RAWQ = cycle_rawq<T, K, PRE, POST, SM, AR>();
    using V = typename VecOf<T>::type;
    constexpr int W = VecOf<T>::W;
    constexpr int CW = W / 2;
    constexpr int ETOP = POST ? 1 : 0;
    constexpr int EBOT = POST == 1 ? 2 : (POST == 2 ? 1 : 0);
    const V Z = vzero((V*)nullptr);
    const long ccol = col / 2;
    const bool cld = ld && (ccol + CW < cpitch);
    V lev[K + 1][3];      // level windows in rotating slots; level K only when POST
    V bw[BL ? 1 : K + 1]; // bw[j] = rhs row y-1-j (in the wave's LDS ring instead when BL)
#pragma unroll
    for (int j = 0; j <= K; ++j) { lev[j][0] = Z; lev[j][1] = Z; lev[j][2] = Z; }
#pragma unroll
    for (int j = 0; j < (BL ? 1 : K + 1); ++j) bw[j] = Z;
    if constexpr (BL) {
#pragma unroll
        for (int q = 0; q < kBRing; ++q) ring_put(ring, q, Z);
    }
    V rawq[RAWQ ? kBRing : 1];                       // [step phase]: b on its way from the second ring to the residual stage
#pragma unroll
    for (int q = 0; q < (RAWQ ? kBRing : 1); ++q) rawq[q] = Z;
    if constexpr (RAWQ) {
#pragma unroll
        for (int q = 0; q < kRawLds; ++q) ring_put(ring2, q, Z);
    }
    CycleState<T, CW> cs;
    cs.acc = 0.0;
#pragma unroll
    for (int k = 0; k < CW; ++k) cs.ct[k] = cs.tc[k] = cs.em[k] = cs.mc[k] = (T)0;
    CycleArgs ca;
    ca.cpitch = cpitch; ca.NC = N / 2; ca.r0 = r0; ca.r1 = r1; ca.zero_in = zero_in; ca.win = win;
    const int y0 = r0 - K - ETOP;
    ca.y_end = r1 + K + EBOT + (cycle_skew<T, K, PRE, POST, SM, AR>() > 0 ? 1 : 0);      // exclusive end of the steps that matter
    // rounded up to whole rotations; the deep (BL) bodies run whole kBRing-step trips with no exit in
    // between - a branch-free trip is what lets the compiler keep several rows in flight - and the
    // launcher picks the chunk height so that nothing (or one step) is wasted
    constexpr int kRound = BL ? kBRing : trip_steps<T>();
    const int steps = (ca.y_end - y0 + kRound - 1) / kRound * kRound;
    constexpr int PFD = cycle_pfd<BL, POST, EDGE>();
    V nin[PFD][kPfStages], nbn[PFD][kPfStages];    // [step phase mod PFD][queue position]
    constexpr int CPFD = cycle_cpfd<T, BL, EDGE, POST>();
    PreFetch<T, CW> pe[CPFD];                       // [step phase mod CPFD]
#pragma unroll
    for (int q = 0; q < CPFD; ++q) {
#pragma unroll
        for (int k = 0; k <= CW; ++k) { pe[q].a[k] = (T)0; pe[q].b[k] = (T)0; }
    }
#pragma unroll
    for (int q = 0; q < PFD; ++q)
        cycle_loads<T, EDGE, ZIN>(nin[q][0], nbn[q][0], y0 + q, pv, pb, pitch, N, ca.y_end, ld, ca.zero_in, ca.win, fo);
    if (PRE) {
#pragma unroll
        for (int q = 0; q < CPFD; ++q) coarse_loads<T, EDGE>(pe[q], y0 + q, coarse_e, cpitch, ccol, N, cld, ca.win, fo);
    }
#define MGX_CSTEP(RP, Y) cycle_step<T, K, PRE, POST, SM, EDGE, RP, BL, ZIN, AR>(lev, bw, ring, ring2, rawq, nin[(RP) % PFD], nbn[(RP) % PFD], pe[(RP) % CPFD], cs, Y, pv, pb, po, coarse_e, coarse_b, coarse_zero, wgt, pitch, col, ccol, N, ca, ld, cld, st, c0, c1, fo)
    if constexpr (BL) {
        // kBRing steps per trip so that every ring slot is a compile-time offset
#define MGX_CTRIP(Y) do { MGX_CSTEP(0, Y); MGX_CSTEP(1, Y + 1); MGX_CSTEP(2, Y + 2); MGX_CSTEP(3, Y + 3); MGX_CSTEP(4, Y + 4); \
                          MGX_CSTEP(5, Y + 5); MGX_CSTEP(6, Y + 6); MGX_CSTEP(7, Y + 7); MGX_CSTEP(8, Y + 8); MGX_CSTEP(9, Y + 9); \
                          MGX_CSTEP(10, Y + 10); MGX_CSTEP(11, Y + 11); } while (0)
#if MGX_PRIO_ALT
        // The SIMD's issue arbiter serves the OLDER of its two waves first: in the wave trace of a two-round launch one
        // wave of every pair finishes after 159 us, its partner after 235 us, the second round starts staggered by that
        // much and the launch ends with a fifth of its span at less than half occupancy.  The two waves of a SIMD (hardware
        // wave slots of opposite parity) therefore take turns at the higher priority, one 12-step trip each.
        unsigned hw_id;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw_id));
        int turn = (int)(hw_id & 1u);
        for (int y = y0; y < y0 + steps; y += kBRing) {
            if (turn & 1) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(0);
            turn ^= 1;
            MGX_CTRIP(y);
        }
        __builtin_amdgcn_s_setprio(0);
#else
        for (int y = y0; y < y0 + steps; y += kBRing) MGX_CTRIP(y);
#endif
#undef MGX_CTRIP
    } else if constexpr (trip_steps<T>() == 12) {
        for (int y = y0; y < y0 + steps; y += 12) {
            MGX_CSTEP(0, y); MGX_CSTEP(1, y + 1); MGX_CSTEP(2, y + 2);
            MGX_CSTEP(3, y + 3); MGX_CSTEP(4, y + 4); MGX_CSTEP(5, y + 5);
            MGX_CSTEP(6, y + 6); MGX_CSTEP(7, y + 7); MGX_CSTEP(8, y + 8);
            MGX_CSTEP(9, y + 9); MGX_CSTEP(10, y + 10); MGX_CSTEP(11, y + 11);
        }
    } else {
        for (int y = y0; y < y0 + steps; y += 3) {
            MGX_CSTEP(0, y); MGX_CSTEP(1, y + 1); MGX_CSTEP(2, y + 2);
        }
    }
#undef MGX_CSTEP
    return cs.acc;
}

// deep passes (rhs ring in LDS): ask for two workgroups per CU = two waves per SIMD, i.e. at most 256
// registers - without it the compiler settles for one wave per SIMD and accumulator-register spills
// workgroups per CU the compiler must make room for (= waves per SIMD): the deep passes with their rhs window in
// LDS ask for two (at most 256 registers).  Three (168 registers) was measured in round 3 for the 10-level
// pre-smoothing kernels, which need 173-177: they spill 60-116 B per lane into the interior loop and the pass takes
// 0.53 instead of 0.40 ms at 8192^2 (profiles/r03_experiments.md).
template <typename T, int K, int PRE, int POST, int SM, int AR> constexpr int cycle_waves()
{
    return cycle_b_in_lds<T, K, POST, SM>() ? 2 : 1;
}
template <typename T, int K, int PRE, int POST, int SM = 0, int AR = 0>
__global__ void __launch_bounds__(kBlock, (cycle_waves<T, K, PRE, POST, SM, AR>()))
k_jacobi_cycle(const T* __restrict__ vin, const T* __restrict__ rhs, T* __restrict__ vout,
               const T* __restrict__ coarse_e,                       // PRE
               T* __restrict__ coarse_b, T* __restrict__ coarse_zero, T wgt,   // POST == 1
               double* __restrict__ partial,                          // POST == 2
               int N, long pitch, long cpitch, int row_lo, int row_hi, int R, int strips, int chunks, int Re, int chunks_e,
               int row_last0, int Rl, int RB, int n_tall, int n_short, int Rf, T c0, T c1, int zero_in, CycleWin win)
{
    constexpr int W = VecOf<T>::W;
    constexpr int XC = cycle_halo_cols<K, POST>();
    constexpr int HL = (XC + W - 1) / W;
    constexpr int OUT = kWave - 2 * HL;
    constexpr int ETOP = POST ? 1 : 0;
    constexpr int EBOT = POST == 1 ? 2 : (POST == 2 ? 1 : 0);
    __shared__ double wsum[kWavesPerBlock];
    // the waves' rhs rings (deep passes only: cycle_b_in_lds)
    constexpr bool BL = cycle_b_in_lds<T, K, POST, SM>();
    __shared__ typename LdsVec<T>::v bring[BL ? kWavesPerBlock * kBRing * kWave : 1];
    constexpr bool RAWQ = cycle_rawq<T, K, PRE, POST, SM, AR>();
    __shared__ typename LdsVec<T>::v braw[RAWQ ? kWavesPerBlock * kRawLds * kWave : 1];
    const CTile t = cycle_tile(strips, chunks, chunks_e, R, Re, row_lo, row_hi, row_last0, Rl, RB, n_tall, n_short, Rf);
    double acc = 0.0;
#ifdef MGX_WAVE_TRACE
    const long long trace_t0 = wall_clock64();
#endif
    if (t.active) {
        const int lane = threadIdx.x & 63;
        const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
        lds_vec_ptr<T> ring = (lds_vec_ptr<T>)&bring[BL ? (wv * kBRing * kWave + lane) : 0];
        lds_vec_ptr<T> ring2 = (lds_vec_ptr<T>)&braw[RAWQ ? (wv * kRawLds * kWave + lane) : 0];
        const int vx0 = t.strip * OUT - HL;
        const int vx = vx0 + lane;
        const long col = (long)vx * W;
        const bool ld = (vx >= 0) && (col + W <= pitch);
        const bool st = (lane >= HL) && (lane < kWave - HL) && (vx < N / W);
        const int r0 = t.r0, r1 = t.r1;
        // everything the unpredicated body touches (rows r0-K-ETOP-1 .. r1+K+EBOT+1+kPrefetch:
        // rotation rounding plus the prefetched rows; one vector beyond the first and last lane;
        // the matching coarse rows/columns) strictly inside the grid and inside the window
        // (the deep bodies round their step count up to whole kBRing-step trips)
        constexpr int kRound = BL ? kBRing : trip_steps<T>();
        const int y_first = r0 - K - ETOP - 1;
        constexpr int SKD = cycle_skew<T, K, PRE, POST, SM, AR>() > 0 ? 1 : 0;            // the skewed chain's extra step
        const int y_lastp = (r0 - K - ETOP) + ((r1 + K + EBOT + SKD) - (r0 - K - ETOP) + kRound - 1) / kRound * kRound + kPrefetchMax;
        // The unpredicated body is safe when the rows its STORED values depend on - input rows y0 = r0 - K - ETOP ..
        // y_end - 1 = r1 + K + EBOT - 1 - are unknown rows that exist: the rows it touches beyond them (the rhs row of
        // the first step, the rounding of the step count to whole trips, the prefetched rows) are loaded from the
        // nearest row that exists (cycle_loads clamps the row number, two scalar instructions) and only feed values
        // that are never stored.  The first and last chunks of a slab of a row-decomposed grid, whose halo rows are
        // exactly the cone, therefore run the interior body too.
        const int y0c = r0 - K - ETOP, y1c = r1 + K + EBOT - 1;
        bool interior = (vx0 >= 1) && ((long)(vx0 + kWave + 1) * W < N) &&
                        (y0c > 0) && (y1c < N) && (y0c >= win.row_first) && (y1c <= win.row_last);
        if (PRE) interior = interior && (y0c >> 1) >= win.crow_first && ((y1c + 1) >> 1) <= win.crow_last;
        interior = interior && !(PRE != 0 && zero_in);
        // both bodies store (and the edge body loads) through buffer descriptors that start at the first
        // row this wave can touch, with 32-bit offsets relative to it
        FastOut fo;
        fo.rb = max(y_first, win.row_first);
        fo.crb = max(y_first >> 1, win.crow_first);
        fo.pitch_bytes = (unsigned)(pitch * (long)sizeof(T));
        fo.cpitch_bytes = (unsigned)(cpitch * (long)sizeof(T));
        const unsigned f_bytes = rsrc_bytes(fo.rb, win.row_last, fo.pitch_bytes);
        const unsigned c_bytes = rsrc_bytes(fo.crb, win.crow_last, fo.cpitch_bytes);
        const long f_at = (long)fo.rb * pitch, c_at = (long)fo.crb * cpitch;
        fo.out = make_rsrc(vout + f_at, f_bytes);
        fo.in = make_rsrc(vin + f_at, f_bytes);
        fo.rhs = make_rsrc(rhs + f_at, f_bytes);
        fo.ce = make_rsrc(PRE ? coarse_e + c_at : nullptr, (PRE && coarse_e) ? c_bytes : 0u);
        fo.cb = make_rsrc((POST == 1 && coarse_b) ? coarse_b + c_at : nullptr, (POST == 1 && coarse_b) ? c_bytes : 0u);
        fo.cz = make_rsrc((POST == 1 && coarse_zero) ? coarse_zero + c_at : nullptr, (POST == 1 && coarse_zero) ? c_bytes : 0u);   // empty: every store dropped
        fo.lane_off = (unsigned)(col * (long)sizeof(T));
        fo.clane_off = (unsigned)((col / 2) * (long)sizeof(T));
        if (interior) {
            if (PRE == 0 && zero_in)
                acc = cycle_body<T, K, PRE, POST, SM, false, PRE == 0, AR>(vin + col, rhs + col, vout + col, coarse_e, coarse_b, coarse_zero, wgt,
                                                                      pitch, cpitch, col, N, r0, r1, true, st, c0, c1, true, win, ring, ring2, fo);
            else
                acc = cycle_body<T, K, PRE, POST, SM, false, false, AR>(vin + col, rhs + col, vout + col, coarse_e, coarse_b, coarse_zero, wgt,
                                                                    pitch, cpitch, col, N, r0, r1, true, st, c0, c1, false, win, ring, ring2, fo);
        } else {
            acc = cycle_body<T, K, PRE, POST, SM, true, false, AR>(vin + col, rhs + col, vout + col, coarse_e, coarse_b, coarse_zero, wgt,
                                                        pitch, cpitch, col, N, r0, r1, ld, st, c0, c1, zero_in != 0, win, ring, ring2, fo);
        }
    }
#ifdef MGX_WAVE_TRACE
    if ((threadIdx.x & 63) == 0) {
        const int w = blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
        if (w < (1 << 16)) {
            unsigned hw, xcc;
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
            // (HW_ID bits 16-19, the workgroup id within the CU, are not used by tools/wave_trace.py: the XCC id goes there)
            g_wave_trace[w] = WaveTrace{trace_t0, wall_clock64(), t.active ? t.strip : -1, t.r0, t.r1,
                                        (int)((hw & 0xFFF0FFFFu) | ((xcc & 0xFu) << 16))};
        }
        if (w == 0) g_wave_trace_n = gridDim.x * kWavesPerBlock;
    }
#endif
    if (POST == 2) {
        for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, kWave);
        if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = acc;
        __syncthreads();
        if (threadIdx.x == 0) {
            double sum = 0.0;
            for (int w = 0; w < kWavesPerBlock; ++w) sum += wsum[w];
            partial[blockIdx.x] = sum;
        }
    }
}

// =============================================================================
// k_tile_smooth: a whole smoothing block on a SMALL level in one launch.
//
// Levels up to ~1024^2 are latency-bound for the marching kernels: a wave walks
// its R + 2K rows one after the other and there are fewer waves than SIMDs.
// Here a workgroup takes a tile and performs ALL `levels` smoother levels on it
// in lockstep, with the tile in REGISTERS and LDS carrying the halo rows between
// waves:
//   * the workgroup's array is 56 rows x 64 columns of nodes = the output tile
//     plus a halo of He = levels (+2 / +1 for a folded restriction / norm) nodes
//     on every side; halo nodes are recomputed redundantly (level j is valid on
//     the array minus j nodes per side), so tiles never talk to each other;
//   * lane = column, wave w owns the 14-row band [14 w, 14 w + 14): u and b of
//     the band stay in 2 x 14 registers per lane for the whole kernel;
//   * x-neighbours come from the adjacent lanes by DPP, the y-neighbours of the
//     band's first / last row from the adjacent wave through a double-buffered
//     LDS halo (one barrier per level).
// Same arithmetic in the same order as every other kernel: bit-identical.
//   SM 0: weighted Jacobi.  SM 1: red-black Gauss-Seidel, level j updates colour
//   (j-1)&1 in place (its neighbours all have the other colour).
//   PRE 1: v + P e while loading (PS:620-624).  POST 1: residual of the result,
//   restricted into the coarse right-hand side, coarse guess zeroed (PS:604-613).
//   POST 2: sum (b - A v)^2 -> partial[block].
// Whole grids only (rows/cols 0..N exist, coarse node I on fine node 2I).
// =============================================================================
constexpr int kTileBand = 14;                       // array rows per wave
constexpr int kTileSY = kWavesPerBlock * kTileBand; // 56 array rows per workgroup
constexpr int kTileSX = kWave;                      // 64 array columns

template <int POST> constexpr int tile_extra() { return POST == 1 ? 2 : (POST == 2 ? 1 : 0); }

template <typename T, int SM, int PRE, int POST, int AR = 0>
__global__ void __launch_bounds__(kBlock)
k_tile_smooth(const T* __restrict__ vin, const T* __restrict__ rhs, T* __restrict__ vout,
              const T* __restrict__ coarse_e, T* __restrict__ coarse_b, T* __restrict__ coarse_zero, T wgt,
              double* __restrict__ partial, int N, long pitch, long cpitch, int levels, T c0, T c1,
              int tiles_x, int zero_in, int row_lo, int row_hi, CycleWin win)
{
    constexpr int RW = kTileBand;
    __shared__ T edge[2][kWavesPerBlock][2][kWave];     // [buffer][wave][first / last row][lane]
    __shared__ double wsum[kWavesPerBlock];
    const int He = levels + tile_extra<POST>();
    const int TH = kTileSY - 2 * He, TW = kTileSX - 2 * He;    // output tile
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int ty = blockIdx.x / tiles_x, tx = blockIdx.x - ty * tiles_x;
    const int y0 = wv * RW;                         // array row of this wave's register row 0
    // rows [row_lo, row_hi) are updated (whole grid: 1 .. N-1); on a slab of a row-decomposed grid the base pointers
    // are moved back so that GLOBAL row numbers index them, and `win` says which rows (and coarse rows) exist
    const int gy0 = row_lo + ty * TH - He + y0;     // its global node row
    const int gx = 1 + tx * TW - He + lane;         // this lane's global node column
    const bool colunk = gx > 0 && gx < N;
    const int NC = N / 2;

    // ---- load the band (with the correction, PRE): straight-line, every load in flight at once ----
    T u[RW], b[RW];
    unsigned rowmask = 0;                           // bit i: register row i is a row of unknowns
#pragma unroll
    for (int i = 0; i < RW; ++i) {
        const int gy = gy0 + i;
        const bool rowok = gy > 0 && gy < N && gy >= win.row_first && gy <= win.row_last;     // wave-uniform
        if (rowok) rowmask |= 1u << i;
        const bool unk = colunk && rowok;
        const long at = (long)gy * pitch + gx;
        u[i] = (unk && !zero_in) ? vin[at] : (T)0;
        b[i] = unk ? rhs[at] : (T)0;
    }
    if (PRE) {
        const bool codd = (gx & 1) != 0;
#pragma unroll
        for (int i = 0; i < RW; ++i) {
            const int gy = gy0 + i;
            const bool rodd = (gy & 1) != 0;                               // wave-uniform
            const bool unk = colunk && ((rowmask >> i) & 1u) && (gy >> 1) >= win.crow_first && (gy >> 1) + (rodd ? 1 : 0) <= win.crow_last;
            // the four coarse nodes around (gy, gx); all inside the coarse grid when (gy, gx) is an unknown
            const T* e = coarse_e + (long)(gy >> 1) * cpitch + (gx >> 1);
            const T a00 = unk ? e[0] : (T)0;
            const T a01 = (unk && codd) ? e[1] : (T)0;
            const T a10 = (unk && rodd) ? e[cpitch] : (T)0;
            const T a11 = (unk && rodd && codd) ? e[cpitch + 1] : (T)0;
            const T ev = codd ? (T)0.5 * (a00 + a01) : a00;                          // even row
            const T od = codd ? (T)0.25 * (((a00 + a10) + a01) + a11) : (T)0.5 * (a00 + a10);
            const T add = rodd ? od : ev;
            u[i] = unk ? u[i] + add : u[i];
        }
    }

    // ---- the smoother levels, in lockstep over the workgroup ---------------------------
    int buf = 0;
    auto halo_rows = [&](const T (&a)[RW], T& up, T& dn) {
        edge[buf][wv][0][lane] = a[0];
        edge[buf][wv][1][lane] = a[RW - 1];
        __syncthreads();
        up = wv > 0 ? edge[buf][wv - 1][1][lane] : (T)0;
        dn = wv < kWavesPerBlock - 1 ? edge[buf][wv + 1][0][lane] : (T)0;
        buf ^= 1;       // the next exchange writes the other buffer; its barrier orders the re-use of this one
    };
    for (int j = 1; j <= levels; ++j) {
        T up, dn;
        halo_rows(u, up, dn);
        const bool colact = colunk && lane >= j && lane < kTileSX - j;
        // rows of this band inside level j's region: i in [j - y0, 56 - j - y0)
        const int ilo = max(j - y0, 0), ihi = min(kTileSY - j - y0, RW);
        const unsigned act = ihi > ilo ? (rowmask & (((1u << (ihi - ilo)) - 1u) << ilo)) : 0u;
        // straight-line over the 14 rows (inactive rows are computed and dropped): the rows are
        // independent, so their dependent add chains interleave
        T prev = up;                                // old value of the row above
#pragma unroll
        for (int i = 0; i < RW; ++i) {
            const T cur = u[i];
            const T below = (i == RW - 1) ? dn : u[i + 1];
            const T l = from_left(cur), r = from_right(cur);
            const bool rowact = (act >> i) & 1u;                                // wave-uniform
            if (SM == 0) {
                const T o = jac_pt<AR>(c0, cur, c1 * b[i], c1, nbr(prev, l, r, below));
                u[i] = (colact && rowact) ? o : cur;
            } else {
                const T o = (T)0.25 * (b[i] + nbr(prev, l, r, below));
                u[i] = (colact && rowact && ((gy0 + i + gx) & 1) == ((j - 1) & 1)) ? o : cur;
            }
            prev = cur;
        }
    }
    // u is the result on array rows / columns [levels, 56 - levels) x [levels, 64 - levels)

    // ---- store the tile -------------------------------------------------------------------------
    const bool colout = lane >= He && lane < He + TW && gx < N;
#pragma unroll
    for (int i = 0; i < RW; ++i) {
        const int y = y0 + i, gy = gy0 + i;
        if (y >= He && y < He + TH && gy < N && gy >= row_lo && gy < row_hi && colout) vout[(long)gy * pitch + gx] = u[i];
    }

    if (POST != 0) {
        // residual of the result: r = b - A u on the tile (+- 1 node for the restriction)
        T up, dn;
        halo_rows(u, up, dn);
        T res[RW];
        double acc = 0.0;
        T prev = up;
#pragma unroll
        for (int i = 0; i < RW; ++i) {
            const T cur = u[i];
            const T below = (i == RW - 1) ? dn : u[i + 1];
            const int y = y0 + i, gy = gy0 + i;
            const T l = from_left(cur), r = from_right(cur);
            T rr = (T)0;
            if (colunk && gy > 0 && gy < N) rr = b[i] - (-nbr(prev, l, r, below) + (T)4 * cur);
            res[i] = rr;
            if (POST == 2 && y >= He && y < He + TH && gy >= row_lo && gy < row_hi && colout) acc += (double)rr * (double)rr;
            prev = cur;
        }
        if (POST == 2) {
            for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, kWave);
            if (lane == 0) wsum[wv] = acc;
            __syncthreads();
            if (threadIdx.x == 0) {
                double sum = 0.0;
                for (int w2 = 0; w2 < kWavesPerBlock; ++w2) sum += wsum[w2];
                partial[blockIdx.x] = sum;
            }
        }
        if (POST == 1) {
            T rup, rdn;
            halo_rows(res, rup, rdn);
            const int J = gx >> 1;
            const bool cst = colout && (gx & 1) == 0 && J >= 1 && J < NC;
#pragma unroll
            for (int i = 0; i < RW; ++i) {
                const T c = res[i];
                const T n = (i == 0) ? rup : res[i - 1];
                const T s2 = (i == RW - 1) ? rdn : res[i + 1];
                const int y = y0 + i, gy = gy0 + i;
                // PS:539-542 order: ((nw+ne)+sw)+se + 2*(((w+e)+n)+s) + 4*c
                T corners = from_left(n) + from_right(n);
                corners = corners + from_left(s2);
                corners = corners + from_right(s2);
                T edges = from_left(c) + from_right(c);
                edges = edges + n;
                edges = edges + s2;
                const T o = wgt * ((corners + (T)2 * edges) + (T)4 * c);
                if (cst && y >= He && y < He + TH && (gy & 1) == 0 && gy < N && gy >= row_lo && gy < row_hi &&
                    (gy >> 1) >= win.emit_lo && (gy >> 1) < win.emit_hi) {          // a coarse row of this tile
                    const long at = (long)(gy >> 1) * cpitch + J;
                    coarse_b[at] = o;
                    if (coarse_zero) coarse_zero[at] = (T)0;
                }
            }
        }
    }
}

// mixed precision (config 5): u64 += scale * (double) e32, rows [row_lo,row_hi)
static __global__ void __launch_bounds__(kBlock)
k_axpy_f32_to_f64(double* __restrict__ u, const float* __restrict__ e, double scale, int N, long pitch, long epitch,
                  int row_lo, int row_hi, int R, int strips, int chunks, int assign)
{
    const Tile t = wave_tile(strips, chunks);
    if (!t.active) return;
    const Cols c = lane_cols<2>(t.strip, N, pitch);
    const int r0 = row_lo + t.chunk * R;
    const int r1 = min(r0 + R, row_hi);
    for (int r = r0; r < r1; ++r) {
        if (c.st) {
            const float2 ev = *reinterpret_cast<const float2*>(e + (long)r * epitch + c.col);
            double2* pu = reinterpret_cast<double2*>(u + (long)r * pitch + c.col);
            double2 o = assign ? make_double2(0.0, 0.0) : *pu;
            o.x = o.x + scale * (double)ev.x;
            o.y = o.y + scale * (double)ev.y;
            if (c.col == 0) o.x = 0.0;
            *pu = o;
        }
    }
}

// mixed precision, the two fp64 passes of a cycle in one: unew = u + scale * (double) e
// (k_axpy_f32_to_f64) and, of that new iterate, r = b - A unew: sum r^2 -> partial[] and
// (float)(r * inv_scale) -> r32 (k_residual<double, 2>).  Out of place: a chunk reads its
// neighbours' edge rows of u, which an in-place update would race with.  Same expressions,
// same launch geometry and summation order as the two kernels it replaces: same bits.
// Algorithmic bytes per point: 8 (u) + 4 (e) + 8 (b) + 8 (unew) + 4 (r32) = 32 (separately: 40).
static __global__ void __launch_bounds__(kBlock)
k_update_residual(const double* __restrict__ u, const float* __restrict__ e, const double* __restrict__ rhs,
                  double* __restrict__ unew, float* __restrict__ r32, double scale, double inv_scale,
                  double* __restrict__ partial, int N, long pitch, long epitch, int row_lo, int row_hi,
                  int R, int strips, int chunks)
{
    __shared__ double wsum[kWavesPerBlock];
    const Tile t = wave_tile(strips, chunks);
    double acc = 0.0;
    if (t.active) {
        const Cols c = lane_cols<2>(t.strip, N, pitch);
        const int r0 = row_lo + t.chunk * R;
        const int r1 = min(r0 + R, row_hi);
        // updated row r (rows 0 and N, column 0 and the padding hold zeros in u and e and stay zero)
        auto updated = [&](int r) {
            double2 o = make_double2(0.0, 0.0);
            if (c.ld) {
                const double2 uv = *reinterpret_cast<const double2*>(u + (long)r * pitch + c.col);
                const float2 ev = *reinterpret_cast<const float2*>(e + (long)r * epitch + c.col);
                o.x = uv.x + scale * (double)ev.x;
                o.y = uv.y + scale * (double)ev.y;
                if (c.col == 0) o.x = 0.0;
            }
            return o;
        };
        double2 up = updated(r0 - 1);
        double2 cur = updated(r0);
        for (int r = r0; r < r1; ++r) {
            const double2 dn = updated(r + 1);
            const double2 bb = vload<double2>(rhs + (long)r * pitch + c.col, c.ld);
            double2 o = residual_vec(up, cur, dn, bb);
            mask_cols(o, c.col, N);
            if (c.st) {
                *reinterpret_cast<double2*>(unew + (long)r * pitch + c.col) = cur;
                acc += (double)o.x * (double)o.x + (double)o.y * (double)o.y;
                *reinterpret_cast<float2*>(r32 + (long)r * epitch + c.col) =
                    make_float2((float)(o.x * inv_scale), (float)(o.y * inv_scale));
            }
            up = cur; cur = dn;
        }
    }
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, kWave);
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        double s = 0.0;
        for (int w = 0; w < kWavesPerBlock; ++w) s += wsum[w];
        partial[blockIdx.x] = s;
    }
}

// f64 grid -> f32 grid with scaling (mixed FMG right-hand side)
static __global__ void __launch_bounds__(kBlock)
k_scale_f64_to_f32(float* __restrict__ out, const double* __restrict__ in, double inv_scale, int N, long pitch_in, long pitch_out,
                   int row_lo, int row_hi, int R, int strips, int chunks)
{
    const Tile t = wave_tile(strips, chunks);
    if (!t.active) return;
    const Cols c = lane_cols<2>(t.strip, N, pitch_in);
    const int r0 = row_lo + t.chunk * R;
    const int r1 = min(r0 + R, row_hi);
    for (int r = r0; r < r1; ++r) {
        if (c.st) {
            double2 x = *reinterpret_cast<const double2*>(in + (long)r * pitch_in + c.col);
            if (c.col == 0) x.x = 0.0;
            *reinterpret_cast<float2*>(out + (long)r * pitch_out + c.col) =
                make_float2((float)(x.x * inv_scale), (float)(x.y * inv_scale));
        }
    }
}

// =============================================================================
// launch geometry
// =============================================================================
struct Launch { int R, strips, chunks, blocks; };

// rows: number of rows to process; W: elements per vector; rows_per_chunk <= 0 = auto
inline Launch make_launch(int N, int W, int rows, int rows_per_chunk)
{
    Launch L;
    const int nvec_out = N / W;
    L.strips = (nvec_out + kOutLanes - 1) / kOutLanes;
    if (L.strips < 1) L.strips = 1;
    int R = rows_per_chunk;
    if (R <= 0) {
        // short chunks measured best on MI355X (tools/microbench): many more
        // waves than the chip holds, so the tail is short; the halo-row
        // re-reads are L2 hits.
        R = rows / 128;
        if (R < 4) R = 4;
        if (R > 8) R = 8;
    }
    L.R = R;
    L.chunks = (rows + R - 1) / R;
    if (L.chunks < 1) L.chunks = 1;
    const long waves = (long)L.strips * L.chunks;
    long blocks = (waves + kWavesPerBlock - 1) / kWavesPerBlock;
    blocks = (blocks + 7) / 8 * 8;
    L.blocks = (int)blocks;
    return L;
}

} // namespace mgx
