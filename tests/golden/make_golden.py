"""Generates tests/golden/*.npz|json from the CPU oracle (oracle/mg_oracle.c).

These fixtures are NOT reference outputs: the reference has no tests or golden
vectors and cannot be built in this image (SYCL/oneMKL/Eigen absent), so parity
with it is unpinned.  They freeze the oracle's behaviour so that (a) a change
to the oracle is visible in review and (b) the GPU tests have a committed
expectation that does not depend on the oracle being rebuilt on the GPU box.

Run from the repo root:  python tests/golden/make_golden.py
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import pyoracle as po  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))


def operators(level, seed):
    n = (1 << level) - 1
    nc = (1 << (level - 1)) - 1
    rng = np.random.default_rng(seed)
    d = {}
    v = rng.uniform(-1, 1, (n, n))
    f = rng.uniform(-1, 1, (n, n))
    e = rng.uniform(-1, 1, (nc, nc))
    d["v"], d["f"], d["e"] = v, f, e
    for name, dt in (("f64", np.float64), ("f32", np.float32)):
        vv, ff, ee = v.astype(dt), f.astype(dt), e.astype(dt)
        d[f"jacobi3_{name}"] = po.jacobi(vv, ff, 3)
        d[f"rbgs2_{name}"] = po.rbgs(vv, ff, 2)
        d[f"residual_{name}"] = po.residual(vv, ff)
        d[f"restrict_{name}"] = po.restrict(ff)
        d[f"restrict_fw16_{name}"] = po.restrict(ff, po.RESTRICT_FW16)
        d[f"resrestrict_{name}"] = po.restrict(po.residual(vv, ff))
        d[f"prolong_{name}"] = po.prolong(ee)
        d[f"prolong_add_{name}"] = po.prolong_add(vv, ee)
    return d


HISTORY_CASES = {
    # BASELINE config 1: 256^2, 3-level V-cycle, weighted Jacobi (reference's V(10,10), omega 2/3)
    "c1_L8_3level_jacobi_v1010": dict(finest_level=8, coarsest_level=6, mu1=10, mu2=10, schedule=0),
    "c1_L8_3level_jacobi_v21": dict(finest_level=8, coarsest_level=6, mu1=2, mu2=1, schedule=0),
    # config 2 shape (6 levels, Jacobi) at a size the oracle finishes in seconds
    "c2_L9_6level_jacobi_v21": dict(finest_level=9, coarsest_level=4, mu1=2, mu2=1, schedule=0),
    # config 3 shape: red-black Gauss-Seidel
    "c3_L9_rbgs_v21": dict(finest_level=9, coarsest_level=5, mu1=2, mu2=1, schedule=0, smoother=1),
    # config 5 shape: FMG + mixed precision
    "c5_L9_fmg_mixed": dict(finest_level=9, coarsest_level=6, mu0=0, mu1=2, mu2=1, schedule=1, dtype=2),
    "c5_L9_fmg_f64": dict(finest_level=9, coarsest_level=6, mu0=0, mu1=2, mu2=1, schedule=1, dtype=1),
    # the reference's literal bottom (D8) and weights (D4), for completeness
    "d8_L8_smooth_bottom": dict(finest_level=8, coarsest_level=6, mu1=2, mu2=1, schedule=0, bottom=1),
    "d4_L8_fw16": dict(finest_level=8, coarsest_level=6, mu1=2, mu2=1, schedule=0, restrict_mode=1),
    "f32_L8_v21": dict(finest_level=8, coarsest_level=6, mu1=2, mu2=1, schedule=0, dtype=0),
}


def histories():
    out = {}
    for name, cfg in HISTORY_CASES.items():
        L = cfg["finest_level"]
        for rhs in ("constant", "sine_random_guess"):
            if rhs == "constant":
                b, u0 = po.rhs_constant(L), None
            else:
                b, u0 = po.rhs_sine(L), po.fill_uniform(((1 << L) - 1,) * 2, 12345)
            # float hierarchies: the oracle's sine-transform bottom mode (the device's direct method
            # in the device's operation order), so both sides round the same fp64 bottom solution
            ocfg = dict(cfg)
            if ocfg.get("dtype", 1) != 1 and ocfg.get("bottom", 0) == 0:
                ocfg["bottom"] = po.BOTTOM_DST
            s = po.Solver(**ocfg)
            u, h = s.solve(b, u0, tol=1e-8, max_cycles=20)
            n = u.shape[0]
            out[f"{name}/{rhs}"] = dict(cfg=cfg, history=[float(x) for x in h],
                                        u_centre=float(u[n // 2, n // 2]), u_absmax=float(np.abs(u).max()),
                                        u_sum=float(u.sum()))
    return out


if __name__ == "__main__":
    np.savez_compressed(os.path.join(OUT, "operators_L5.npz"), **operators(5, 2026))
    np.savez_compressed(os.path.join(OUT, "operators_L6.npz"), **operators(6, 2027))
    with open(os.path.join(OUT, "histories.json"), "w") as fh:
        json.dump(histories(), fh, indent=1)
    print("golden fixtures written to", OUT)
