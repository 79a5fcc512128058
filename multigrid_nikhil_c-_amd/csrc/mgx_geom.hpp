// mgx_geom.hpp — which rows and columns each wave of a k_jacobi_cycle launch works on.
//
// Plain C++ (no HIP types): the launcher (mgx_launch.hpp) chooses a geometry with cycle_geom_pick, the kernel
// (mgx_kernels.hpp) decodes its tile with cycle_tile_at, and tests/test_geom.py compiles the same two functions with g++
// and checks on the CPU that every row of every strip is covered exactly once.
//
// A tile = one wave = one strip of columns (the wave's 64 vectors minus halo lanes) x a chunk of rows [r0, r1).
//   strips 1 .. S-2 ("interior strips"): the first chunk Re rows, middle chunks, the last TWO chunks Rl rows each
//                    - the first and last two run (almost certainly) the edge body, ~1.3 x an interior row step, and
//                    are shorter;
//   strips 0 and S-1: chunks_e chunks of Re rows (always the edge body).
// Tiles are dealt to the 8 XCDs an eighth of every class each, x fastest (neighbouring strips share halo columns in
// the XCD's L2), and every XCD starts with its edge-class tiles.
//
// Middle chunks, two forms:
//   uniform (n_tall == 0): chunks - 3 chunks of R rows.
//   paired  (n_tall > 0; launches of ONE round of workgroups only): n_tall chunks of R rows followed by n_short chunks
//     of RB < R rows.  The wave trace of a one-round launch (profiles/r03_experiments.md 21) shows that the two waves of a
//     SIMD do not share it evenly: the wave of the workgroup dispatched first (block index b < 256 = one per CU; its
//     partner is block b + 256) takes 0.75 us per row step, the other one 1.0-1.17, and a wave left alone 0.67 - the SIMD
//     does 2.18 steps/us with two waves and 1.48 with one.  With equal chunks the early wave ends after three quarters of
//     the launch and the late one finishes alone.  Here the workgroups dispatched first (the first 32 per XCD) take the
//     edge-class and the tall tiles, the later ones the short tiles, sized so that both end together.
#pragma once

#if defined(__HIPCC__)
#define MGX_GEOM_HD __host__ __device__ __forceinline__
#else
#define MGX_GEOM_HD inline
#endif

namespace mgx {

constexpr int kGeomWavesPerBlock = 4;
constexpr int kGeomCusPerXcd = 32;                 // MI355X: 256 CUs in 8 XCDs
constexpr int kGeomResidentBlocks = 512;           // two workgroups of the deep passes per CU

struct CycleGeom {
    int R, Re, Rl;              // rows of a middle (tall) chunk, of a chunk of the two edge strips, of each of the last two chunks
    int Rf;                     // rows of the first chunk of an interior strip (Re where the range starts at a boundary, else R)
    int chunks, chunks_e;       // chunks of an interior strip (first + middle + last two), of an edge strip
    int row_last0;              // first row of the last-but-one chunk; 0: uniform tiles of R = Re rows everywhere
    int RB, n_tall, n_short;    // paired form: rows of a short chunk, tall and short chunks per interior strip (n_tall = 0: uniform)
    long waves;
    int blocks;
};

struct CTile { int strip, r0, r1; bool active; };

MGX_GEOM_HD int geom_min(int a, int b) { return a < b ? a : b; }

// the tile of wave `wave` of block `block` in a grid of `grid` blocks (a multiple of 8)
MGX_GEOM_HD CTile cycle_tile_at(int block, int wave, int grid, int strips, int chunks, int chunks_e, int R, int Re, int row_lo,
                                int row_hi, int row_last0, int Rl, int RB, int n_tall, int n_short, int Rf)
{
    const int per_xcd = grid >> 3;
    const int xcd = block & 7;
    const int nth = block >> 3;
    const int si = strips - 2;
    CTile t;
    t.strip = 0; t.r0 = 0; t.r1 = 0; t.active = false;
    if (row_last0 != 0) {
        // E: the first and the last two chunks of every interior strip, every chunk of the two edge strips
        const int n_edge = 3 * si + 2 * chunks_e;
        int e = -1, m = -1, s = -1;                 // index in the E list / the middle (tall) list / the short list
        if (n_tall > 0) {
            // paired: the workgroups dispatched first (one per CU) take E and tall tiles, the later ones short tiles
            const int n_f = n_edge + n_tall * si, n_s = n_short * si;
            const int pf = (n_f + 7) >> 3, ps = (n_s + 7) >> 3;
            if (nth < kGeomCusPerXcd) {
                const int lw = nth * kGeomWavesPerBlock + wave;
                const int f = xcd * pf + lw;
                if (lw < pf && f < n_f) { if (f < n_edge) e = f; else m = f - n_edge; }
            } else {
                const int lw = (nth - kGeomCusPerXcd) * kGeomWavesPerBlock + wave;
                if (lw < ps && xcd * ps + lw < n_s) s = xcd * ps + lw;
            }
        } else {
            const int n_mid = (chunks - 3) * si;
            const int pe = (n_edge + 7) >> 3, pm = (n_mid + 7) >> 3;
            const int lw = nth * kGeomWavesPerBlock + wave;
            if (lw < pe) { if (xcd * pe + lw < n_edge) e = xcd * pe + lw; }
            else if (lw - pe < pm && xcd * pm + (lw - pe) < n_mid) m = xcd * pm + (lw - pe);
        }
        if (e >= 0) {
            t.active = true;
            if (e < 3 * si) {
                const int which = e / si;                                 // 0: first chunk, 1 / 2: the last two
                t.strip = 1 + (e - which * si);
                t.r0 = which == 0 ? row_lo : (which == 1 ? row_last0 : row_last0 + Rl);
                t.r1 = which == 0 ? row_lo + Rf : (which == 1 ? row_last0 + Rl : row_hi);
            } else {
                const int k = e - 3 * si;
                const int chunk = k >> 1;
                t.strip = (k & 1) ? strips - 1 : 0;
                t.r0 = row_lo + chunk * Re;
                t.r1 = geom_min(t.r0 + Re, row_hi);
            }
        } else if (m >= 0) {
            const int chunk = m / si;
            t.active = true;
            t.strip = 1 + (m - chunk * si);
            t.r0 = row_lo + Rf + chunk * R;
            t.r1 = geom_min(t.r0 + R, row_last0);
        } else if (s >= 0) {
            const int chunk = s / si;
            t.active = true;
            t.strip = 1 + (s - chunk * si);
            t.r0 = row_lo + Rf + n_tall * R + chunk * RB;
            t.r1 = geom_min(t.r0 + RB, row_last0);
        }
        t.active = t.active && t.r0 < t.r1;
        return t;
    }
    // uniform tiles: interior strips first (x fastest, a contiguous range per XCD, the upper four XCDs walking
    // theirs backwards: the grid's last chunk row - boundary waves - is then the first thing XCD 7 starts), then the
    // two edge strips
    const int b = xcd * per_xcd + (xcd >= 4 ? per_xcd - 1 - nth : nth);
    const long g = (long)b * kGeomWavesPerBlock + wave;
    const long n_int = si > 0 ? (long)chunks * si : 0;
    if (g < n_int) {
        const int chunk = (int)(g / si);
        t.strip = 1 + (int)(g - (long)chunk * si);
        t.r0 = row_lo + chunk * R;
        t.r1 = geom_min(t.r0 + R, row_hi);
        t.active = t.r0 < t.r1;
    } else {
        const long e = g - n_int;
        int chunk;
        if (si > 0) { chunk = (int)(e >> 1); t.strip = (e & 1) ? strips - 1 : 0; }
        else { chunk = (int)(e / strips); t.strip = (int)(e - (long)chunk * strips); }
        t.r0 = row_lo + chunk * Re;
        t.r1 = geom_min(t.r0 + Re, row_hi);
        t.active = chunk < chunks_e && t.r0 < row_hi;
    }
    return t;
}

// ---- host side: choosing a geometry -------------------------------------------------------------------------------

// the next chunk height >= R (in steps of `step`) whose row steps, R + extra, fill whole loop trips (or all but one step)
inline int trip_rows(int R, int extra, int trip, int step)
{
    int best = -1;
    for (int r = R; r < R + 2 * trip; r += step) {
        const int m = (r + extra) % trip;
        if (m == 0) return r;
        if (m == trip - 1 && best < 0) best = r;
    }
    return best < 0 ? R : best;
}

// height of the edge tiles for interior tiles R rows high: (Re + extra) / (R + extra) ~ 1 - pct/100, the inverse of what
// an edge step costs relative to an interior one, in whole loop trips
inline int edge_rows_pct(int R, int extra, int trip, int pct)
{
    const double frac = 0.01 * (double)pct;
    const int k = (int)(frac * (double)(R + extra) / (double)trip + 0.5);
    const int Re = R - k * trip;
    return Re >= trip ? Re : (R >= 2 * trip ? trip : R);
}

struct GeomKnobs {
    int edge_short = 1;         // MGX_EDGE_SHORT
    int edge_pct = 23;          // MGX_EDGE_PCT: flat from 15 to 36 % at 8192^2, 4096^2 and 2048^2
    int last_pct = 38;          // MGX_LAST_PCT: the last two chunk rows (~1.5 x an interior row step)
    int min_chunk = 16;         // MGX_MIN_CHUNK
    int min_rounds = 1;         // MGX_MIN_ROUNDS
    int min_rounds_rows = 1024; // MGX_MIN_ROUNDS_ROWS
    int pair = 1;               // MGX_PAIR: paired chunk heights in one-round launches
    int pair_ratio = 130;       // MGX_PAIR_RATIO: row steps of a tall chunk, in percent of a short one's
    int pair_max_rows = 640;    // MGX_PAIR_MAX_ROWS: tallest chunk the paired form may use
    int pair_min_rows = 150;    // MGX_PAIR_MIN_ROWS: ranges whose short chunks would be lower keep the uniform rule
};

inline int geom_blocks(const CycleGeom& g, int strips)
{
    int blocks = (int)(((g.waves + kGeomWavesPerBlock - 1) / kGeomWavesPerBlock + 7) / 8 * 8);
    if (g.row_last0 != 0) {
        const int si = strips - 2;
        const int n_edge = 3 * si + 2 * g.chunks_e;
        if (g.n_tall > 0) {
            const int ps = (g.n_short * si + 7) / 8;
            blocks = 8 * (kGeomCusPerXcd + (ps + kGeomWavesPerBlock - 1) / kGeomWavesPerBlock);
        } else {
            const int pe = (n_edge + 7) / 8, pm = ((g.chunks - 3) * si + 7) / 8;
            blocks = 8 * ((pe + pm + kGeomWavesPerBlock - 1) / kGeomWavesPerBlock);
        }
    }
    return blocks;
}

// uniform middle chunks of R rows; chunks of the two edge strips Re; first chunk of an interior strip Rf, its last two Rl
// (callers pass Rf = Re and a shortened Rl where that end of the range is a boundary of the grid or of the rows that
// exist - the waves there run the edge body - and Rf = R / Rl = R where it is not)
inline CycleGeom cycle_geom(int row_lo, int row_hi, int strips, int R, int Re, int Rl, int Rf = -1)
{
    const int rows = row_hi - row_lo;
    if (Rf < 0) Rf = Re;
    CycleGeom g;
    g.R = R; g.Re = Re; g.Rl = Rl; g.Rf = Rf; g.RB = 0; g.n_tall = 0; g.n_short = 0;
    g.chunks_e = (rows + Re - 1) / Re;
    if ((Re >= R && Rf >= R && Rl >= R) || rows <= Rf + 2 * Rl + 2 || strips < 3) {
        // uniform tiles (of the edge height when the range is only a few edge tiles high)
        g.R = (Re >= R) ? R : Re; g.Re = g.R; g.Rl = g.R; g.Rf = g.R;
        g.row_last0 = 0;
        g.chunks = g.chunks_e = (rows + g.R - 1) / g.R;
    } else {
        // first chunk Rf rows; the last two Rl rows each at the end (the very last one row less when the
        // parity of the range asks for it: every chunk starts on a row of row_lo's parity, and one row MORE
        // could cost a whole loop trip)
        const int last = row_hi - Rl + ((row_hi - Rl - row_lo) & 1);
        g.row_last0 = last - Rl;
        g.chunks = 3 + (g.row_last0 - (row_lo + Rf) + R - 1) / R;
    }
    g.waves = strips > 2 ? (long)g.chunks * (strips - 2) + 2L * g.chunks_e : (long)g.chunks_e * strips;
    g.blocks = geom_blocks(g, strips);
    return g;
}

// paired form for ONE round of workgroups; returns false when none fits (extra = row steps of a chunk beyond its rows)
inline bool cycle_geom_paired(int row_lo, int row_hi, int strips, int extra, int trip, const GeomKnobs& kn, bool top_edge, bool bot_edge,
                              CycleGeom* out)
{
    const int rows = row_hi - row_lo;
    const int si = strips - 2;
    if (si < 1) return false;
    const int slots = kGeomCusPerXcd * kGeomWavesPerBlock;        // waves per XCD in each half of the round: 128
    for (int rb = trip_rows(kn.min_chunk, extra, trip, 2); rb <= kn.pair_max_rows; rb += trip) {
        // tall chunk: pair_ratio % of the short one's row steps, in whole trips
        const int steps_b = (rb + extra + trip - 1) / trip * trip;
        const int steps_a = (int)(((long)steps_b * kn.pair_ratio / 100 + trip - 1) / trip) * trip;
        const int ra = (steps_a - extra) & ~1;                        // even; ra + extra fills whole trips (or all but one step)
        if (ra > kn.pair_max_rows) break;
        if (ra <= rb) continue;
        const int re = edge_rows_pct(ra, extra, trip, kn.edge_pct), rl = bot_edge ? edge_rows_pct(ra, extra, trip, kn.last_pct) : ra;
        const int rf = top_edge ? re : ra;
        if (rows <= rf + 2 * rl + 2 + rb) continue;
        const int chunks_e = (rows + re - 1) / re;
        const int n_edge = 3 * si + 2 * chunks_e;
        const int last = row_hi - rl + ((row_hi - rl - row_lo) & 1);
        const int row_last0 = last - rl;
        const int mid = row_last0 - (row_lo + rf);                   // rows the tall and short chunks must cover
        if (mid < rb) continue;
        // most tiles that fit: an eighth of each list per XCD, 128 waves per XCD and half
        const int nt_max = (8 * slots - 7 - n_edge) / si, ns_max = (8 * slots - 7) / si;   // (- 7: the per-XCD rounding)
        if (nt_max < 1 || ns_max < 1) continue;
        if ((long)nt_max * ra + (long)ns_max * rb < mid) continue;   // taller chunks needed
        if (rb < kn.pair_min_rows) return false;                      // small ranges: the uniform rule
        // the rows split between the classes in proportion to what each could take (all slots in use when the fit is
        // tight), at least one short chunk
        int nt = (int)((double)mid * ((double)nt_max * ra / ((double)nt_max * ra + (double)ns_max * rb)) / ra + 0.5);
        if (nt > nt_max) nt = nt_max;
        while (nt > 0 && (long)nt * ra > mid - rb) --nt;
        int ns = nt > 0 ? (mid - nt * ra + rb - 1) / rb : ns_max + 1;
        while (ns > ns_max && nt < nt_max && (long)(nt + 1) * ra <= mid - rb) { ++nt; ns = (mid - nt * ra + rb - 1) / rb; }
        if (nt < 1 || ns > ns_max) continue;
        const int pf = (n_edge + nt * si + 7) / 8, ps = (ns * si + 7) / 8;
        if (pf > slots || ps > slots) continue;
        CycleGeom g;
        g.R = ra; g.Re = re; g.Rl = rl; g.Rf = rf; g.RB = rb; g.n_tall = nt; g.n_short = ns;
        g.chunks_e = chunks_e; g.row_last0 = row_last0; g.chunks = 3 + nt + ns;
        g.waves = (long)g.chunks * si + 2L * chunks_e;
        g.blocks = geom_blocks(g, strips);
        if (g.blocks > kGeomResidentBlocks) continue;
        *out = g;
        return true;
    }
    return false;
}

// what launch_cycle_k uses.  R > 0: uniform chunks of about R rows (MGX_FUSE_ROWS or the shallow passes' rule);
// auto_rows (deep double passes on big grids): the fewest rounds of resident workgroups with chunks of at most ~200
// rows, and in that many rounds the shortest chunks that fit; one round: the paired form when it exists
inline CycleGeom cycle_geom_pick(int row_lo, int row_hi, int strips, int extra, int trip, int R, bool auto_rows, bool deep,
                                 const GeomKnobs& kn, bool top_edge = true, bool bot_edge = true)
{
    if (R & 1) ++R;
    R = trip_rows(R, extra, trip, 2);
    CycleGeom g = cycle_geom(row_lo, row_hi, strips, R, R, R);
    if (!deep) return g;
    auto shaped = [&](int r) {
        return cycle_geom(row_lo, row_hi, strips, r, kn.edge_short ? edge_rows_pct(r, extra, trip, kn.edge_pct) : r,
                          (kn.edge_short && bot_edge) ? edge_rows_pct(r, extra, trip, kn.last_pct) : r,
                          (kn.edge_short && top_edge) ? edge_rows_pct(r, extra, trip, kn.edge_pct) : r);
    };
    if (!auto_rows) return kn.edge_short ? shaped(R) : g;
    if (kn.pair && kn.edge_short && strips >= 3) {
        // ONE round with paired chunk heights, when the range is small enough for it (pair_max_rows)
        CycleGeom p;
        if (cycle_geom_paired(row_lo, row_hi, strips, extra, trip, kn, top_edge, bot_edge, &p)) return p;
    }
    bool found = false;
    for (int m = (row_hi - row_lo >= kn.min_rounds_rows ? kn.min_rounds : 1); m <= 64 && !found; ++m) {
        for (int r = trip_rows(kn.min_chunk, extra, trip, 2); r <= 204; r += trip) {
            const CycleGeom c = shaped(r);
            // (blocks, not only waves: the per-XCD rounding of the two tile classes can add a workgroup or two, and
            // eight workgroups too many are a second round of their own)
            if (c.waves <= 2048L * m && c.blocks <= kGeomResidentBlocks * m) { g = c; found = true; break; }
        }
    }
    return g;
}

} // namespace mgx
