"""Per-GPU compute budget of the multi-GPU driver, measured on ONE GPU: P slabs of the 16384^2 grid all on
device 0 (mgx_config.n_gpus = P, devices = [0] * P) run one after the other, so (time per cycle) / P is what
one GPU of a P-GPU job computes per cycle - its slab passes plus ONE copy of the replicated levels - with the
halo exchanges as same-device copies.  What a real node adds: xGMI latency of the exchanges, the all-gather
at the cut level and the all-reduce of the norm."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge

pkg = ge.load_package()
L = int(sys.argv[1]) if len(sys.argv) > 1 else 14
arith = sys.argv[2] if len(sys.argv) > 2 else "fma"          # what bench.py runs
kw = dict(finest_level=L, coarsest_level=7, mu0=0, mu1=10, mu2=10, schedule=pkg.SCHEDULE_V,
          arith=pkg.ARITH_FMA if arith == "fma" else pkg.ARITH_SEPARATE)
res = {}
for P in (1, 2, 4, 8):
    extra = {} if P == 1 else dict(n_gpus=P, devices=[0] * P)
    with pkg.Multigrid(**kw, **extra) as mg:
        mg.fill_rhs(1, 0.0)
        mg.fill_guess_random(12345)
        mg.solve(tol=0.0, max_cycles=3)
        best = 1e9
        for rep in range(3):
            mg.synchronize()
            t0 = time.perf_counter()
            mg.solve(tol=0.0, max_cycles=5)
            mg.synchronize()
            best = min(best, (time.perf_counter() - t0) / 5 * 1e3)
        ex = mg.exchanges() if P > 1 else 0
    res[P] = best
    print(f"L{L} {arith} P={P}: {best:.3f} ms per cycle for all slabs, {best / P:.3f} ms per GPU-equivalent"
          f" (x{res[1] / (best / P):.2f} of one GPU's {res[1]:.3f} ms)", flush=True)
