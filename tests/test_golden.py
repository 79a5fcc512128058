"""The oracle must keep reproducing the committed fixtures (tests/golden,
written by tests/golden/make_golden.py from the oracle itself; not reference
outputs -- parity with the reference is unpinned)."""
import json
import os

import numpy as np
import pytest

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.mark.parametrize("level", [5, 6])
def test_oracle_reproduces_operator_fixtures(po, level):
    g = np.load(os.path.join(GOLD, f"operators_L{level}.npz"))
    for name, dt in (("f64", np.float64), ("f32", np.float32)):
        v, f, e = g["v"].astype(dt), g["f"].astype(dt), g["e"].astype(dt)
        assert np.array_equal(po.jacobi(v, f, 3), g[f"jacobi3_{name}"])
        assert np.array_equal(po.rbgs(v, f, 2), g[f"rbgs2_{name}"])
        assert np.array_equal(po.residual(v, f), g[f"residual_{name}"])
        assert np.array_equal(po.restrict(f), g[f"restrict_{name}"])
        assert np.array_equal(po.restrict(f, po.RESTRICT_FW16), g[f"restrict_fw16_{name}"])
        assert np.array_equal(po.prolong(e), g[f"prolong_{name}"])
        assert np.array_equal(po.prolong_add(v, e), g[f"prolong_add_{name}"])


def test_oracle_reproduces_history_fixtures(po):
    gold = json.load(open(os.path.join(GOLD, "histories.json")))
    for key, rec in gold.items():
        cfg = rec["cfg"]
        L = cfg["finest_level"]
        if key.endswith("constant"):
            b, u0 = po.rhs_constant(L), None
        else:
            b, u0 = po.rhs_sine(L), po.fill_uniform(((1 << L) - 1,) * 2, 12345)
        ocfg = dict(cfg)
        if ocfg.get("dtype", 1) != 1 and ocfg.get("bottom", 0) == 0:
            ocfg["bottom"] = po.BOTTOM_DST        # float hierarchies: the device's bottom method (make_golden.py)
        u, h = po.Solver(**ocfg).solve(b, u0, tol=1e-8, max_cycles=20)
        assert len(h) == len(rec["history"]), key
        assert np.allclose(h, rec["history"], rtol=1e-12, atol=0), key
        assert abs(u[u.shape[0] // 2, u.shape[0] // 2] - rec["u_centre"]) <= 1e-13 * max(1.0, abs(rec["u_centre"])), key
