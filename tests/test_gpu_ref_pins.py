"""The HIP kernels, through the C-ABI, against data the REFERENCE itself computed
(tests/golden/ref_ps.npz; see tests/test_ref_pins.py for how it was made): k_prolong<float>
== interpolation2d (PS:337-425) bit for bit, mgx_fill_rhs == -globalforcefunction (PS:283-335,
defect D1's sign), and the kernels' operator == the matrix globalstiffenssmatrix assembles."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_ps.npz")


@pytest.fixture(scope="module")
def ref():
    return np.load(GOLD)


@pytest.mark.parametrize("tile", ["tiles", "marching"])
@pytest.mark.parametrize("nc", [3, 7, 31, 63])
def test_device_prolongation_is_bit_equal_to_the_reference(pkg, ref, nc, tile, monkeypatch):
    if tile == "marching":
        monkeypatch.setenv("MGX_TILE_MAX_N", "0")
    level = int(np.log2(nc + 1)) + 1                 # fine level: n = 2 nc + 1
    e, want = ref[f"interp_in_{nc}"], ref[f"interp_out_{nc}"]
    with pkg.Multigrid(finest_level=level, coarsest_level=level - 1, dtype=pkg.DTYPE_F32, bottom=pkg.BOTTOM_SMOOTH) as mg:
        got = mg.interpolation2d(level, e)           # k_prolong<float, false>
        assert got.dtype == np.float32 and np.array_equal(got, want)
        # the correction add of PS:620-624 on top of it: v + P e, one float add per point
        v = np.random.default_rng(5).uniform(-1, 1, want.shape).astype(np.float32)
        assert np.array_equal(mg.interpolation_add(level, v, e), v + want)
    # the same operator in double agrees with the float reference to float rounding
    with pkg.Multigrid(finest_level=level, coarsest_level=level - 1, dtype=pkg.DTYPE_F64, bottom=pkg.BOTTOM_SMOOTH) as mg:
        got = mg.interpolation2d(level, e.astype(np.float64))
        assert np.max(np.abs(got - want)) <= 2 * np.finfo(np.float32).eps * np.max(np.abs(e))


@pytest.mark.parametrize("level", [7, 8, 9, 10])
def test_device_load_vector_equals_the_reference_up_to_d1_sign(pkg, ref, level):
    want = ref[f"force_L{level}"]
    for dt, cast in ((pkg.DTYPE_F32, np.float32), (pkg.DTYPE_F64, np.float64)):
        with pkg.Multigrid(finest_level=level, coarsest_level=min(7, level), dtype=dt) as mg:
            mg.fill_rhs(0, 4.0)                      # mgx_fill_rhs kind 0 = the reference's load vector (PS:123 f = 4)
            got = mg.get_level(level, pkg.VEC_B)
            assert got.dtype == cast and np.array_equal(got, -want.astype(cast))


def test_device_operator_is_the_matrix_the_reference_assembles(pkg, ref):
    """residual kernel (PS:604-607) on integer data == b - A v with A from the reference's COO
    triplets (globalstiffenssmatrix PS:200-281; sign per D1), 15^2 unknowns = level 4"""
    nodes, n, level = 17, 15, 4
    A = np.zeros((n * n, n * n))
    for nm in ("lu", "d"):
        np.add.at(A, (ref[f"coo{nodes}_rows_{nm}"], ref[f"coo{nodes}_cols_{nm}"]), ref[f"coo{nodes}_vals_{nm}"].astype(np.float64))
    rng = np.random.default_rng(11)
    v = rng.integers(-8, 9, (n, n)).astype(np.float64)
    b = rng.integers(-8, 9, (n, n)).astype(np.float64)
    want = b - (-A @ v.ravel()).reshape(n, n)
    for dt, cast in ((pkg.DTYPE_F32, np.float32), (pkg.DTYPE_F64, np.float64)):
        with pkg.Multigrid(finest_level=level, coarsest_level=level, dtype=dt, bottom=pkg.BOTTOM_SMOOTH) as mg:
            assert np.array_equal(mg.residual(level, v.astype(cast), b.astype(cast)), want.astype(cast))
