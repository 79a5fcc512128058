"""kernel trace target: P = 8 slabs of 16384^2 on one device, a few cycles (see tools/slab_budget.py)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge

pkg = ge.load_package()
P = int(sys.argv[1]) if len(sys.argv) > 1 else 8
with pkg.Multigrid(finest_level=14, coarsest_level=7, mu0=0, mu1=10, mu2=10, schedule=pkg.SCHEDULE_V, n_gpus=P, devices=[0] * P, arith=pkg.ARITH_FMA) as mg:
    mg.fill_rhs(1, 0.0)
    mg.fill_guess_random(12345)
    mg.solve(tol=0.0, max_cycles=4)
    mg.synchronize()
