"""ms per mgx_solve loop body with and without the phase profile (eager launches + HIP events vs
hipGraph replay), same handle configuration as bench.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge

pkg = ge.load_package()
for L in (13, 10):
    for prof in (1, 0):
        with pkg.Multigrid(finest_level=L, coarsest_level=7, mu0=0, mu1=10, mu2=10, schedule=pkg.SCHEDULE_V, profile=prof) as mg:
            mg.fill_rhs(1, 0.0)
            mg.fill_guess_random(12345)
            mg.solve(tol=0.0, max_cycles=5)
            best = 1e9
            for rep in range(3):
                mg.synchronize()
                t0 = time.perf_counter()
                mg.solve(tol=0.0, max_cycles=20)
                mg.synchronize()
                best = min(best, (time.perf_counter() - t0) / 20 * 1e3)
            print(f"L{L} profile={prof}: {best:.4f} ms per step", flush=True)
