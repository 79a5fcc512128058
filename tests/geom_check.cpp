// CPU check of the tile map of k_jacobi_cycle (csrc/mgx_geom.hpp): for whole grids and row slabs, the geometry the
// launcher would pick covers every row of every strip exactly once, chunks start on rows of row_lo's parity, the grid is
// a multiple of 8 workgroups, and a paired (one-round) geometry never exceeds the resident workgroups.
// Built and run by tests/test_geom.py;  argv[1] = "v" prints every case.
#include "mgx_geom.hpp"
#include <cstdio>
#include <vector>
#include <cstdlib>
using namespace mgx;
int main(int argc, char** argv)
{
    int fails = 0;
    struct Case { int N, lo, hi, K, POST, W; };
    std::vector<Case> cases;
    for (int L = 8; L <= 14; ++L) for (int post = 0; post < 3; ++post) { int N = 1 << L; cases.push_back({N, 1, N, 10, post, 2}); cases.push_back({N, 1, N, 8, post, 2}); }
    for (int P : {2, 4, 8}) for (int L = 11; L <= 14; ++L) for (int post = 0; post < 3; ++post) { int N = 1 << L; int rows = N / P; for (int r = 0; r < P; ++r) { int lo = r * rows + 1, hi = (r + 1) * rows + (r == P - 1 ? 0 : 1); if (r == P - 1) hi = N; cases.push_back({N, lo, std::min(hi, N), 10, post, 2}); } }
    // float passes: four columns per lane, half the strips (the launcher lowers the paired form's threshold to 100 rows there)
    for (int L = 11; L <= 14; ++L) for (int post = 0; post < 3; ++post) { int N = 1 << L; cases.push_back({N, 1, N, 10, post, 4}); }
    for (int P : {2, 8}) for (int L = 12; L <= 14; ++L) { int N = 1 << L; int rows = N / P; for (int r = 0; r < P; ++r) cases.push_back({N, r * rows + 1, r == P - 1 ? N : (r + 1) * rows + 1, 10, 1, 4}); }
    // FUZZ=<n>: n more cases with random levels, row ranges (odd first row), depths and stages
    if (getenv("FUZZ")) {
        unsigned long long st = 88172645463325252ull;
        auto rnd = [&](int n) { st ^= st << 13; st ^= st >> 7; st ^= st << 17; return (int)(st % (unsigned long long)n); };
        for (int i = 0; i < atoi(getenv("FUZZ")); ++i) {
            const int L = 11 + rnd(4), N = 1 << L;
            int lo = 1 + 2 * rnd(N / 2 - 40), hi = lo + 24 + rnd(N - lo - 24);
            if (hi > N) hi = N;
            if (rnd(4) == 0) { lo = 1; }
            if (rnd(4) == 0) { hi = N; }
            cases.push_back({N, lo, hi, rnd(2) ? 10 : 8, rnd(3), rnd(3) ? 2 : 4});
        }
    }
    GeomKnobs kn;
    if (getenv("MINCHUNK")) kn.min_chunk = atoi(getenv("MINCHUNK"));
    if (getenv("PAIRMIN")) kn.pair_min_rows = atoi(getenv("PAIRMIN"));
    if (getenv("EDGEPCT")) kn.edge_pct = atoi(getenv("EDGEPCT"));
    if (getenv("LASTPCT")) kn.last_pct = atoi(getenv("LASTPCT"));
    if (getenv("PAIR")) kn.pair = atoi(getenv("PAIR"));
    if (getenv("RATIO")) kn.pair_ratio = atoi(getenv("RATIO"));
    for (const Case& c : cases) {
        const int halo = c.K + (c.POST == 1 ? 2 : (c.POST == 2 ? 1 : 0));
        const int OUT = 64 - 2 * ((halo + c.W - 1) / c.W);
        const int strips = (c.N / c.W + OUT - 1) / OUT;
        const int E = c.POST == 1 ? 3 : (c.POST == 2 ? 2 : 0);
        const int extra = 2 * c.K + E, trip = 12;
        if (c.POST == 1 && !(c.lo & 1)) continue;
        // (a slab's inner ends are not edges: its halo rows hold the cone)
        const bool top_edge = c.lo - c.K - 2 < 1, bot_edge = c.hi + c.K + 2 > c.N - 1;
        GeomKnobs kc = kn;
        if (c.W == 4 && kc.pair_min_rows > 100) kc.pair_min_rows = 100;
        if (c.K == 8 && c.W == 4) continue;               // (no deep 8-level float pass: its window is in registers)
        const CycleGeom g = cycle_geom_pick(c.lo, c.hi, strips, extra, trip, 64, true, true, kc, top_edge, bot_edge);
        std::vector<int> cover((size_t)strips * (c.hi - c.lo), 0);
        long active = 0; int maxsteps = 0;
        for (int b = 0; b < g.blocks; ++b) for (int w = 0; w < 4; ++w) {
            const CTile t = cycle_tile_at(b, w, g.blocks, strips, g.chunks, g.chunks_e, g.R, g.Re, c.lo, c.hi, g.row_last0, g.Rl, g.RB, g.n_tall, g.n_short, g.Rf);
            if (!t.active) continue;
            ++active;
            if (t.strip < 0 || t.strip >= strips || t.r0 < c.lo || t.r1 > c.hi || ((t.r0 - c.lo) & 1)) { printf("BAD tile N=%d b=%d w=%d strip=%d r0=%d r1=%d\n", c.N, b, w, t.strip, t.r0, t.r1); ++fails; continue; }
            for (int r = t.r0; r < t.r1; ++r) ++cover[(size_t)t.strip * (c.hi - c.lo) + (r - c.lo)];
            if (t.r1 - t.r0 + extra > maxsteps) maxsteps = t.r1 - t.r0 + extra;
        }
        long bad = 0; for (int v : cover) if (v != 1) ++bad;
        if (g.n_tall > 0 && g.blocks > kGeomResidentBlocks) { printf("PAIRED geometry of %d blocks N=%d rows %d..%d\n", g.blocks, c.N, c.lo, c.hi); ++fails; }
        if (bad || (g.blocks & 7)) { printf("COVERAGE FAIL N=%d rows %d..%d K=%d POST=%d: %ld cells, blocks %d\n", c.N, c.lo, c.hi, c.K, c.POST, bad, g.blocks); ++fails; }
        if (argc > 1) printf("N=%5d rows %5d..%5d K=%2d POST=%d strips %3d: R %3d Re %3d Rf %3d Rl %3d RB %3d tall %2d short %2d chunks %2d/%2d blocks %4d (%.2f rounds) active %ld longest %d steps\n",
               c.N, c.lo, c.hi, c.K, c.POST, strips, g.R, g.Re, g.Rf, g.Rl, g.RB, g.n_tall, g.n_short, g.chunks, g.chunks_e, g.blocks, g.blocks / 512.0, active, maxsteps);
    }
    printf("%s (%zu cases)\n", fails ? "FAILED" : "ok", cases.size());
    return fails != 0;
}
