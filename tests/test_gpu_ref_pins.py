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


def lu_ref(ref, nodes):
    """off-diagonal part of the reference's assembled matrix in the SPD sign (D1): LU = -1 on the four
    neighbours, from the `lu` COO triplets of globalstiffenssmatrix (PS:200-281)"""
    n = nodes - 2
    LU = np.zeros((n * n, n * n))
    np.add.at(LU, (ref[f"coo{nodes}_rows_lu"], ref[f"coo{nodes}_cols_lu"]), ref[f"coo{nodes}_vals_lu"].astype(np.float64))
    return -LU, n


@pytest.mark.parametrize("fma", [0, 1])
def test_device_jacobi_sweep_at_the_reference_weight_against_the_reference_matrix(pkg, ref, fma):
    """A1 pinned on the device: one sweep of PS:125-147 in the reference's own precision (fp32) and weight
    (omega = 2/3, PS:127) on integer data ==  (1 - w) v + (w/4) f + (-w/4)(LU_ref v)  (PS:138-142) with LU_ref taken
    from the matrix the reference itself assembled: within 2 ulp of the largest value (three roundings of the
    scalars, two of the sum), in both arithmetic modes; and in double to double rounding"""
    nodes, level = 17, 4
    LU, n = lu_ref(ref, nodes)
    rng = np.random.default_rng(17)
    v = rng.integers(-8, 9, (n, n)).astype(np.float64)
    f = rng.integers(-8, 9, (n, n)).astype(np.float64)
    for dt, cast in ((pkg.DTYPE_F32, np.float32), (pkg.DTYPE_F64, np.float64)):
        om = float(cast(2.0 / 3.0))                              # the weight as the precision holds it (PS:127 float omega)
        want = (1.0 - om) * v + (om / 4.0) * f + (-om / 4.0) * (LU @ v.ravel()).reshape(n, n)      # exact products and sums in double
        with pkg.Multigrid(finest_level=level, coarsest_level=level, dtype=dt, bottom=pkg.BOTTOM_SMOOTH, arith=fma) as mg:
            got = mg.jacobirelaxation(level, v.astype(cast), f.astype(cast), 1)
        assert np.max(np.abs(got.astype(np.float64) - want)) <= 2 * np.finfo(cast).eps * np.max(np.abs(want))


@pytest.mark.parametrize("nc", [31, 63])
def test_device_restriction_is_the_transpose_of_the_reference_interpolation(pkg, ref, nc):
    """A3 pinned on the device: <R r, e> = <r, P_ref e> with P_ref e computed BY THE REFERENCE (interpolation2d
    PS:337-425, fixture interp_out_*): the consistent restriction (PS:531-546 index pattern, weight 1/4 per D4)
    is the transpose of the reference's own prolongation.  r: integers, so R r is exact."""
    level = int(np.log2(nc + 1)) + 1
    e = ref[f"interp_in_{nc}"].astype(np.float64)
    pe = ref[f"interp_out_{nc}"].astype(np.float64)            # P_ref e, reference-computed
    rng = np.random.default_rng(nc)
    r = rng.integers(-16, 17, pe.shape).astype(np.float64)
    with pkg.Multigrid(finest_level=level, coarsest_level=level - 1, bottom=pkg.BOTTOM_SMOOTH) as mg:
        Rr = mg.restriction2d(level, r)                          # k_restrict<double, false>
    lhs, rhs = float(np.sum(Rr * e)), float(np.sum(r * pe))
    assert abs(lhs - rhs) <= 2 * np.finfo(np.float32).eps * np.sum(np.abs(r * pe))      # P_ref e is fp32-rounded
    # and through the folded pass of the cycle (k_tile_smooth / k_jacobi_cycle POST = 1): zero sweeps are not a
    # cycle, so use the residual form: with v = 0 the residual is b, R(b - A 0) = R b
    with pkg.Multigrid(finest_level=level, coarsest_level=level - 1, bottom=pkg.BOTTOM_SMOOTH) as mg:
        cb, _ = mg.residual_restriction(level, np.zeros_like(r), r)
    assert np.array_equal(cb, Rr)
