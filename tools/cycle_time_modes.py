"""ms per V(10,10) cycle of mgx_solve at 8192^2 fp64 (FMA) with profiling off (the whole cycle + norm replayed from one
hipGraph), and with cfg.profile = 2 (what bench.py times: the finest-level passes eager between events, the rest one graph)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge

pkg = ge.load_package()
for profile in (0, 2, 0, 2):
    with pkg.Multigrid(finest_level=13, coarsest_level=7, mu0=0, mu1=10, mu2=10, schedule=pkg.SCHEDULE_V, arith=pkg.ARITH_FMA,
                       profile=profile) as mg:
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
        print(f"profile={profile}: {best:.4f} ms per cycle (incl. one initial residual norm per solve call / 20)")
