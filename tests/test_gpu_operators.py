"""GPU parity, operator by operator: the HIP kernels behind the C-ABI against
the CPU oracle on the same seeded inputs, and against the committed fixtures.

The kernels are compiled with -ffp-contract=off and perform the oracle's IEEE
operations in the oracle's order, so every grid operator (Jacobi, RB-GS,
residual, restriction, prolongation, correction) is required to be BIT-EXACT in
both precisions.  Only two things are compared to a tolerance: the exact bottom
solve (sine transform on the device, banded Cholesky in the oracle: two
different exact methods, <= 1e-11 relative) and norms (different summation
order, <= 1e-12 relative)."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
F32_ULP = float(np.finfo(np.float32).eps)

DT = {
    "f64": (np.float64, 1, 1e-12),
    "f32": (np.float32, 0, 4 * F32_ULP),
}


def close(a, b, tol, scale=None):
    s = np.max(np.abs(b)) if scale is None else scale
    return np.max(np.abs(a.astype(np.float64) - b.astype(np.float64))) <= tol * max(s, 1e-300)


def exact(a, b):
    assert a.dtype == b.dtype and a.shape == b.shape
    return np.array_equal(a, b)


@pytest.fixture(scope="module", params=["tiles", "marching"])
def mgs(pkg, request):
    """one handle per (dtype, smoother) spanning levels 2..10; once with the LDS tile kernel
    on the small levels (the default), once with the marching kernels everywhere"""
    old = os.environ.get("MGX_TILE_MAX_N")
    if request.param == "marching":
        os.environ["MGX_TILE_MAX_N"] = "0"
    else:
        os.environ.pop("MGX_TILE_MAX_N", None)
    hs = {}
    try:
        for name, (dt, code, _) in DT.items():
            for sm in (0, 1):
                hs[(name, sm)] = pkg.Multigrid(finest_level=10, coarsest_level=2, dtype=code, smoother=sm,
                                               bottom=pkg.BOTTOM_SMOOTH)
    finally:
        if old is None:
            os.environ.pop("MGX_TILE_MAX_N", None)
        else:
            os.environ["MGX_TILE_MAX_N"] = old
    yield hs
    for h in hs.values():
        h.close()


@pytest.mark.parametrize("name", ["f64", "f32"])
@pytest.mark.parametrize("level", [2, 3, 5, 6, 7, 8, 9, 10])
def test_jacobi_matches_oracle(mgs, po, name, level):
    dt, _, tol = DT[name]
    n = (1 << level) - 1
    rng = np.random.default_rng(100 + level)
    v = rng.uniform(-1, 1, (n, n)).astype(dt)
    f = rng.uniform(-1, 1, (n, n)).astype(dt)
    for mu in (1, 2, 5):
        got = mgs[(name, 0)].jacobirelaxation(level, v, f, mu)
        assert exact(got, po.jacobi(v, f, mu)), (name, level, mu)


@pytest.mark.parametrize("name", ["f64", "f32"])
@pytest.mark.parametrize("level", [2, 3, 5, 6, 7, 8, 9, 10])
def test_rbgs_matches_oracle(mgs, po, name, level):
    dt, _, tol = DT[name]
    n = (1 << level) - 1
    rng = np.random.default_rng(200 + level)
    v = rng.uniform(-1, 1, (n, n)).astype(dt)
    f = rng.uniform(-1, 1, (n, n)).astype(dt)
    for mu in (1, 2, 3):
        got = mgs[(name, 1)].jacobirelaxation(level, v, f, mu)
        assert exact(got, po.rbgs(v, f, mu)), (name, level, mu)


@pytest.mark.parametrize("name", ["f64", "f32"])
@pytest.mark.parametrize("level", [2, 3, 6, 7, 9, 10])
def test_residual_matches_oracle(mgs, po, name, level):
    dt, _, tol = DT[name]
    n = (1 << level) - 1
    rng = np.random.default_rng(300 + level)
    v = rng.uniform(-1, 1, (n, n)).astype(dt)
    f = rng.uniform(-1, 1, (n, n)).astype(dt)
    got = mgs[(name, 0)].residual(level, v, f)
    assert exact(got, po.residual(v, f))
    # and its norm
    mg = mgs[(name, 0)]
    mg.set_level(level, 0, v)
    mg.set_level(level, 1, f)
    assert abs(mg.residual_norm(level) - po.norm2(po.residual(v, f))) <= (1e-12 if name == "f64" else 1e-5) * po.norm2(po.residual(v, f))


@pytest.mark.parametrize("name", ["f64", "f32"])
@pytest.mark.parametrize("level", [3, 4, 6, 7, 9, 10])
def test_restriction_and_fused_residual_restriction(mgs, po, name, level):
    dt, _, tol = DT[name]
    n = (1 << level) - 1
    rng = np.random.default_rng(400 + level)
    v = rng.uniform(-1, 1, (n, n)).astype(dt)
    f = rng.uniform(-1, 1, (n, n)).astype(dt)
    mg = mgs[(name, 0)]
    assert exact(mg.restriction2d(level, f), po.restrict(f))
    cb, cu = mg.residual_restriction(level, v, f)
    assert exact(cb, po.restrict(po.residual(v, f)))
    assert np.all(cu == 0)          # PS:613: the coarse guess is zeroed in the same pass


@pytest.mark.parametrize("name", ["f64", "f32"])
@pytest.mark.parametrize("level", [3, 4, 6, 7, 9, 10])
def test_prolongation_matches_oracle(mgs, po, name, level):
    dt, _, tol = DT[name]
    n, nc = (1 << level) - 1, (1 << (level - 1)) - 1
    rng = np.random.default_rng(500 + level)
    v = rng.uniform(-1, 1, (n, n)).astype(dt)
    e = rng.uniform(-1, 1, (nc, nc)).astype(dt)
    mg = mgs[(name, 0)]
    assert exact(mg.interpolation2d(level, e), po.prolong(e))
    assert exact(mg.interpolation_add(level, v, e), po.prolong_add(v, e))


def test_survey_pin_prolongation_of_ones_on_device(mgs):
    p = mgs[("f64", 0)].interpolation2d(3, np.ones((3, 3)))
    assert np.all(p[1:-1, 1:-1] == 1.0) and p[0, 0] == 0.25 and np.all(p[0, 1:-1] == 0.5)


def test_restriction_weight_modes(pkg, po):
    rng = np.random.default_rng(7)
    f = rng.uniform(-1, 1, (63, 63))
    with pkg.Multigrid(finest_level=6, coarsest_level=5, restrict_mode=pkg.RESTRICT_FW16) as mg:
        assert exact(mg.restriction2d(6, f), po.restrict(f, po.RESTRICT_FW16))


@pytest.mark.parametrize("level", [2, 5, 6, 7, 8])
@pytest.mark.parametrize("name", ["f64", "f32"])
def test_exact_bottom_solve(pkg, po, level, name):
    dt, code, _ = DT[name]
    n = (1 << level) - 1
    rng = np.random.default_rng(600 + level)
    u = rng.uniform(-1, 1, (n, n))
    f = (4 * u - (np.pad(u, 1)[:-2, 1:-1] + np.pad(u, 1)[2:, 1:-1] + np.pad(u, 1)[1:-1, :-2] + np.pad(u, 1)[1:-1, 2:]))
    with pkg.Multigrid(finest_level=level, coarsest_level=level, dtype=code) as mg:
        x = mg.bottom_solve(f.astype(dt))
    ref = po.Solver(finest_level=level, coarsest_level=level).bottom_solve(f.astype(dt).astype(np.float64))
    assert close(x, ref, 1e-11 if name == "f64" else 2 * F32_ULP)
    if name == "f64":
        assert close(x, u, 1e-11)


@pytest.mark.parametrize("level", [5, 6])
def test_device_matches_committed_fixtures(mgs, level):
    g = np.load(os.path.join(GOLD, f"operators_L{level}.npz"))
    for name, (dt, _, tol) in DT.items():
        v, f, e = g["v"].astype(dt), g["f"].astype(dt), g["e"].astype(dt)
        mj, mr = mgs[(name, 0)], mgs[(name, 1)]
        assert exact(mj.jacobirelaxation(level, v, f, 3), g[f"jacobi3_{name}"])
        assert exact(mr.jacobirelaxation(level, v, f, 2), g[f"rbgs2_{name}"])
        assert exact(mj.residual(level, v, f), g[f"residual_{name}"])
        assert exact(mj.restriction2d(level, f), g[f"restrict_{name}"])
        assert exact(mj.residual_restriction(level, v, f)[0], g[f"resrestrict_{name}"])
        assert exact(mj.interpolation2d(level, e), g[f"prolong_{name}"])
        assert exact(mj.interpolation_add(level, v, e), g[f"prolong_add_{name}"])


def test_boundary_ring_and_padding_stay_zero(pkg):
    """edge case: data that is largest next to the Dirichlet ring; smoothing must
    treat the ring as exact zeros on every side (PS:188-198, 224)."""
    rng = np.random.default_rng(8)
    n = 31
    v = np.zeros((n, n)); f = np.zeros((n, n))
    v[0, :] = v[-1, :] = v[:, 0] = v[:, -1] = 1.0
    with pkg.Multigrid(finest_level=5, coarsest_level=4) as mg:
        out = mg.jacobirelaxation(5, v, f, 1)
    # corner: (1-w)*1 + (w/4)*(two ring zeros + two ones)
    w = 2.0 / 3.0
    assert abs(out[0, 0] - ((1 - w) + (w / 4) * 2)) < 1e-15
    assert abs(out[0, 5] - ((1 - w) + (w / 4) * 2)) < 1e-15
    assert abs(out[1, 1] - ((w / 4) * 2)) < 1e-15


def test_smoother_is_linear_and_zero_preserving(mgs):
    rng = np.random.default_rng(9)
    n = 255
    v = rng.uniform(-1, 1, (n, n)); f = rng.uniform(-1, 1, (n, n))
    mg = mgs[("f64", 0)]
    a = mg.jacobirelaxation(8, v, f, 2)
    b = mg.jacobirelaxation(8, 2 * v, 2 * f, 2)
    assert np.array_equal(b, 2 * a)           # scaling by 2 is exact in binary
    assert np.all(mg.jacobirelaxation(8, 0 * v, 0 * f, 3) == 0)


@pytest.mark.parametrize("name", ["f64", "f32"])
@pytest.mark.parametrize("kmax", [2, 3, 4, 5, 10])
def test_fused_sweeps_are_bit_identical_to_single_sweeps(pkg, po, name, kmax, monkeypatch):
    """temporal fusion (k_jacobi_fused: K sweeps per pass) must not change a bit"""
    dt, code, _ = DT[name]
    monkeypatch.setenv("MGX_FUSE", str(kmax))
    monkeypatch.setenv("MGX_FUSE_ROWS", "16")
    monkeypatch.setenv("MGX_TILE_MAX_N", "0")          # the marching kernels on every level
    rng = np.random.default_rng(700 + kmax)
    with pkg.Multigrid(finest_level=11, coarsest_level=8, dtype=code, bottom=pkg.BOTTOM_SMOOTH) as mg:
        for level in (8, 9, 10, 11):
            n = (1 << level) - 1
            v = rng.uniform(-1, 1, (n, n)).astype(dt)
            f = rng.uniform(-1, 1, (n, n)).astype(dt)
            # with kmax = 10 the planner uses K = 6, 8, 10 and mixed splits as well
            for mu in sorted({kmax, kmax + 1, 6, 8, 10, 14} if kmax == 10 else {kmax, kmax + 1, 10}):
                assert exact(mg.jacobirelaxation(level, v, f, mu), po.jacobi(v, f, mu)), (name, kmax, level, mu)
    # data largest next to the Dirichlet ring: every level of the fusion must re-zero the ring
    with pkg.Multigrid(finest_level=10, coarsest_level=7, dtype=code, bottom=pkg.BOTTOM_SMOOTH) as mg:
        n = 1023
        v = np.zeros((n, n), dtype=dt); f = np.zeros((n, n), dtype=dt)
        v[0, :] = v[-1, :] = v[:, 0] = v[:, -1] = 1
        f[0, 0] = f[-1, -1] = 3
        assert exact(mg.jacobirelaxation(10, v, f, kmax), po.jacobi(v, f, kmax))


@pytest.mark.parametrize("name", ["f64", "f32"])
@pytest.mark.parametrize("tile_k", [10, 4, 3])
def test_lds_tile_smoother_is_bit_identical(pkg, po, name, tile_k, monkeypatch):
    """k_tile_smooth (all sweeps of a block on 32 x 32 LDS tiles with halos, tile_k levels per
    launch) against the oracle: every level it serves, both smoothers, sweep counts that need
    one, two and three launches, and data concentrated next to the Dirichlet ring"""
    dt, code, _ = DT[name]
    monkeypatch.setenv("MGX_TILE_K", str(tile_k))
    monkeypatch.delenv("MGX_TILE_MAX_N", raising=False)
    rng = np.random.default_rng(900 + tile_k)
    for sm, orc in ((0, po.jacobi), (1, po.rbgs)):
        with pkg.Multigrid(finest_level=10, coarsest_level=2, dtype=code, smoother=sm, bottom=pkg.BOTTOM_SMOOTH) as mg:
            for level in (2, 3, 4, 5, 6, 7, 9, 10):
                n = (1 << level) - 1
                v = rng.uniform(-1, 1, (n, n)).astype(dt)
                f = rng.uniform(-1, 1, (n, n)).astype(dt)
                for mu in ((1, 2, 5, 10, 11, 23) if sm == 0 else (1, 2, 5, 6, 11)):
                    if level >= 9 and mu > 11:
                        continue
                    assert exact(mg.jacobirelaxation(level, v, f, mu), orc(v, f, mu)), (name, sm, tile_k, level, mu)
            n = 1023
            v = np.zeros((n, n), dtype=dt); f = np.zeros((n, n), dtype=dt)
            v[0, :] = v[-1, :] = v[:, 0] = v[:, -1] = 1
            v[31, :] = v[32, :] = v[:, 31] = v[:, 32] = -2          # tile seams
            f[0, 0] = f[-1, -1] = 3
            assert exact(mg.jacobirelaxation(10, v, f, 7), orc(v, f, 7))
