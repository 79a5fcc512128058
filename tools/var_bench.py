#!/usr/bin/env python3
"""Throughput of the general per-level operator path (op = MGX_OPERATOR_STENCIL5, csrc/mgx_var.hpp) on one MI355X:
the sweep of MF:75-96 and the residual of MF:150-153 move 8 sizeof(T) per point (v, b, five coefficient grids in,
one grid out) - HBM-bound single passes.  Prints a markdown table (kept as profiles/r03_var_operators.md):
    python tools/var_bench.py [level] """
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge

pkg = ge.load_package()
L = int(sys.argv[1]) if len(sys.argv) > 1 else 13
n = (1 << L) - 1
x = np.linspace(0.0, 1.0, n + 2)
a = 1.0 + 0.8 * np.sin(3 * np.pi * x)[None, :] * np.cos(2 * np.pi * x)[:, None]
print("| kernel | grid | dtype | ms per pass | algorithmic GB/s (8 sizeof(T) per point) | fraction of the 8 TB/s HBM peak | G updates/s |")
print("|---|---|---|---|---|---|---|")
for dtype, name, es in ((pkg.DTYPE_F64, "f64", 8), (pkg.DTYPE_F32, "f32", 4)):
    with pkg.Multigrid(finest_level=L, coarsest_level=5, mu1=2, mu2=2, schedule=pkg.SCHEDULE_V, op=pkg.OPERATOR_STENCIL5, dtype=dtype,
                       omega=0.8) as mg:
        mg.set_coefficient(a)
        mg.fill_rhs(1, 0.0)
        mg.fill_guess_random(12345)
        mg.time_smoother(5)
        sweeps = 40
        ms = mg.time_smoother(sweeps) / sweeps
        gbs = 8.0 * es * n * n / (ms * 1e-3) / 1e9
        print(f"| k_jacobi_var<{'double' if es == 8 else 'float'}> (MF:75-96) | {1 << L}^2 | {name} | {ms:.4f} | {gbs:.0f} | {gbs / 8000:.3f} | {n * n / (ms * 1e-3) / 1e9:.1f} |", flush=True)
        mg.fill_guess_random(12345)
        t0 = time.perf_counter()
        reps = 20
        for _ in range(reps):
            mg.residual_norm(L)
        dt = (time.perf_counter() - t0) / reps * 1e3
        gbs = 7.0 * es * n * n / (dt * 1e-3) / 1e9
        print(f"| k_residual_var<.., 1> + reduction (MF:150-153, norm only: 7 sizeof(T) per point; wall time incl. the host sync) | {1 << L}^2 | {name} | {dt:.4f} | {gbs:.0f} | {gbs / 8000:.3f} | |", flush=True)
        if dtype == pkg.DTYPE_F64:
            mg.fill_guess_random(12345)
            st, h = mg.solve(tol=1e-8, max_cycles=40)
            print(f"\nV(2,2) solve, full weighting, smooth coefficient (contrast 9), {L - 5 + 1} levels, f64: {st.cycles} cycles to 1e-8, "
                  f"{st.seconds * 1e3:.1f} ms, convergence factors {np.round(h[1:6] / h[:5], 3).tolist()}\n", flush=True)
