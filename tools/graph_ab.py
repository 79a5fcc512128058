#!/usr/bin/env python3
"""Time mgx_solve's loop with and without hipGraph replay (profiling off)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge

pkg = ge.load_package()
for L, C, mu1, mu2, cycles in ((8, 6, 10, 10, 200), (10, 7, 10, 10, 200), (10, 7, 2, 1, 200), (12, 7, 10, 10, 50), (13, 7, 10, 10, 20)):
    row = []
    for graph in ("0", "1", "0", "1"):
        os.environ["MGX_GRAPH"] = graph
        with pkg.Multigrid(finest_level=L, coarsest_level=C, mu1=mu1, mu2=mu2, schedule=0, profile=0) as mg:
            mg.fill_rhs(1, 0.0)
            mg.fill_guess_random(1)
            mg.solve(tol=0.0, max_cycles=5)
            mg.fill_guess_random(1)
            mg.synchronize()
            t0 = time.perf_counter()
            st, h = mg.solve(tol=0.0, max_cycles=cycles)
            dt = time.perf_counter() - t0
            row.append((dt / cycles * 1e3, mg.graphs_cached()))
    print(f"L{L}..{C} V({mu1},{mu2}): " + "  ".join(f"graphs={g}: {ms:.4f} ms/cycle" for (ms, g) in row), flush=True)
