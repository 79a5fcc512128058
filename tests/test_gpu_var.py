"""GPU parity of the general per-level operator path (handles with op = MGX_OPERATOR_STENCIL5; csrc/mgx_var.hpp)
against the oracle's restatement of MF (oracle/mg_oracle_var.inc; its constant-coefficient case is pinned by
the reference's own assembled matrix in tests/test_var_oracle.py): SURVEY §8(f)4.
Bit-exact: the Jacobi splitting, MF's sweep v <- R_omega v + omega D^-1 b, the residual, injection, the
re-discretisation from a nodal coefficient, the dense direct bottom solve.  Residual histories of whole
solves: 1e-10 relative per cycle (north_star)."""
import numpy as np
import pytest

from test_gpu_solve import hist_close
from test_var_oracle import coefficient, ref_csr

pytestmark = pytest.mark.gpu

VAR = 1


@pytest.mark.parametrize("dtype", [1, 0])
@pytest.mark.parametrize("level,kind", [(5, "smooth"), (7, "jump"), (9, "smooth"), (10, "smooth")])
def test_operators_are_bit_identical_to_the_oracle(pkg, po, dtype, level, kind):
    dt = np.float64 if dtype == 1 else np.float32
    n = (1 << level) - 1
    a = coefficient(level, kind)
    coef = [x.astype(dt) for x in po.stencil_from_nodes(a, level, level)]
    rng = np.random.default_rng(level)
    v = rng.uniform(-1, 1, (n, n)).astype(dt)
    b = rng.uniform(-1, 1, (n, n)).astype(dt)
    omega = 0.8
    jac = po.var_build_jacobi(*coef, omega=omega)
    with pkg.Multigrid(finest_level=level, coarsest_level=max(2, level - 2), op=VAR, dtype=dtype, omega=omega, bottom=pkg.BOTTOM_SMOOTH) as mg:
        with pytest.raises(pkg.MgxError, match="not set"):
            mg.smooth(level, 1)
        mg.set_coefficient(a)                                   # every level, re-discretised on the device
        for q in range(5):
            assert np.array_equal(mg.get_stencil(level, q), coef[q]), q
            assert np.array_equal(mg.get_stencil(level, 5 + q), jac[q]), q
        c2 = po.stencil_from_nodes(a, level - 1, level)
        assert np.array_equal(mg.get_stencil(level - 1, 0), c2[0].astype(dt))
        for mu in (1, 4, 7):                                    # MF:75-96
            assert np.array_equal(mg.jacobirelaxation(level, v, b, mu), po.var_jacobi(v, b, mu, omega, jac)), mu
        assert np.array_equal(mg.residual(level, v, b), po.var_residual(v, b, coef))            # MF:150-153
        rn = mg.residual_norm(level)
        r = po.var_residual(v, b, coef).astype(np.float64)
        assert abs(rn - np.sqrt(np.sum(r * r))) <= 1e-12 * rn
        # the same operator given level by level through mgx_set_stencil (MF: the front-end supplies A_sp_dict[level])
        mg.set_stencil(level, *coef)
        assert np.array_equal(mg.jacobirelaxation(level, v, b, 3), po.var_jacobi(v, b, 3, omega, jac))


@pytest.mark.parametrize("dtype", [1, 0])
def test_the_reference_matrix_through_the_device(pkg, po, dtype):
    """constant coefficients: the device's sweep == the oracle's CSR sweep on the reference's own assembled
    matrix (PS:200-281 triplets -> CSR, MF's data layout), bit for bit"""
    from test_var_oracle import jacobi_csr_of, stencil_arrays

    dt = np.float64 if dtype == 1 else np.float32
    indptr, indices, values, n = ref_csr(17)                    # 15 x 15 unknowns = level 4
    rng = np.random.default_rng(4)
    v = rng.standard_normal((n, n)).astype(dt)
    b = rng.standard_normal((n, n)).astype(dt)
    omega = 2.0 / 3.0
    dinv, r_values = jacobi_csr_of(indptr, indices, values, omega, dt)
    want = po.csr_jacobi(v.ravel(), b.ravel(), 5, omega, indptr, indices, r_values, dinv).reshape(n, n)
    with pkg.Multigrid(finest_level=4, coarsest_level=3, op=VAR, dtype=dtype, omega=omega) as mg:
        mg.set_stencil(4, *stencil_arrays(n, dt))
        mg.set_stencil(3, *stencil_arrays(7, dt))
        assert np.array_equal(mg.jacobirelaxation(4, v, b, 5), want)
        av = po.csr_gemv(indptr, indices, values.astype(dt), v.ravel()).reshape(n, n)
        assert np.array_equal(mg.residual(4, v, b), b - av)


@pytest.mark.parametrize("mode", [0, 2, 3])
@pytest.mark.parametrize("dtype", [1, 0])
def test_restriction_modes_on_general_operators(pkg, po, mode, dtype):
    dt = np.float64 if dtype == 1 else np.float32
    level = 8
    n = (1 << level) - 1
    a = coefficient(level, "smooth")
    coef = [x.astype(dt) for x in po.stencil_from_nodes(a, level, level)]
    rng = np.random.default_rng(9)
    v = rng.uniform(-1, 1, (n, n)).astype(dt)
    b = rng.uniform(-1, 1, (n, n)).astype(dt)
    r = po.var_residual(v, b, coef)
    want = po.restrict(r) if mode == 0 else po.restrict_inject(r, 4.0 if mode == 3 else 1.0)
    with pkg.Multigrid(finest_level=level, coarsest_level=5, op=VAR, dtype=dtype, restrict_mode=mode) as mg:
        mg.set_coefficient(a)
        cb, cu = mg.residual_restriction(level, v, b)            # MF:150-156
        assert np.array_equal(cb, want) and not cu.any()
        got = mg.restriction2d(level, b)                         # of a plain vector (FMG right-hand sides, PS:641)
        assert np.array_equal(got, po.restrict(b) if mode == 0 else po.restrict_inject(b, 4.0 if mode == 3 else 1.0))


@pytest.mark.parametrize("mode", [2, 3])
def test_injection_on_the_constant_stencil_path_too(pkg, po, mode):
    """MGX_RESTRICT_INJECT* with the Poisson operator: the cycle leaves the folded (full-weighting) kernels"""
    cfg = dict(finest_level=9, coarsest_level=6, mu1=2, mu2=2, schedule=0, restrict_mode=mode)
    b = po.rhs_sine(9)
    u0 = po.fill_uniform(b.shape, 3)
    with pkg.Multigrid(**cfg) as mg:
        mg.set_rhs(b)
        mg.set_guess(u0)
        st, h = mg.solve(tol=1e-9, max_cycles=12)
        u = mg.get_solution()
    u_ref, h_ref = po.Solver(**cfg).solve(b, u0, tol=1e-9, max_cycles=12)
    assert hist_close(h, h_ref), (h, h_ref)
    assert np.max(np.abs(u - u_ref)) <= 1e-11 * np.max(np.abs(u_ref))


@pytest.mark.parametrize("cfg", [
    dict(finest_level=8, coarsest_level=4, mu1=2, mu2=2, schedule=0),
    dict(finest_level=9, coarsest_level=3, mu1=3, mu2=1, schedule=0, restrict_mode=3),
    dict(finest_level=8, coarsest_level=5, mu0=0, mu1=2, mu2=2, schedule=1),
    dict(finest_level=8, coarsest_level=4, mu1=2, mu2=2, schedule=0, restrict_mode=2),
    dict(finest_level=8, coarsest_level=5, mu1=4, mu2=4, schedule=0, bottom=1),
    dict(finest_level=9, coarsest_level=4, mu1=2, mu2=1, schedule=0, dtype=0),
    dict(finest_level=10, coarsest_level=2, mu1=10, mu2=10, schedule=0),        # MF:85's ten sweeps
])
@pytest.mark.parametrize("kind", ["smooth", "jump"])
def test_histories_match_the_oracle(pkg, po, cfg, kind):
    cfg = dict(cfg, op=VAR, omega=0.8)
    L = cfg["finest_level"]
    a = coefficient(L, kind)
    b = po.rhs_sine(L)
    u0 = None if cfg["schedule"] == 1 else po.fill_uniform(b.shape, 21)
    with pkg.Multigrid(**cfg) as mg:
        mg.set_coefficient(a)
        mg.set_rhs(b)
        if u0 is not None:
            mg.set_guess(u0)
        st, h = mg.solve(tol=1e-9, max_cycles=14)
        u = mg.get_solution()
        graphs = mg.graphs_cached()
    ref = po.Solver(**cfg)
    ref.set_coefficient(a)
    u_ref, h_ref = ref.solve(b, u0, tol=1e-9, max_cycles=14)
    if cfg.get("dtype", 1) == 1:
        assert hist_close(h, h_ref), (h, h_ref)
        assert np.max(np.abs(u - u_ref)) <= 1e-11 * np.max(np.abs(u_ref))
    else:                                                        # float: every operator is bit-exact, the histories equal to rounding of the norm
        assert len(h) == len(h_ref) and np.allclose(h, h_ref, rtol=1e-6, atol=0)
        assert np.array_equal(u.astype(np.float32), u_ref.astype(np.float32))
    assert graphs >= 1                                           # the cycle was replayed from a hipGraph


def test_direct_bottom_solve_of_a_general_operator(pkg, po):
    for L in (3, 4, 5):
        a = coefficient(L, "smooth")
        n = (1 << L) - 1
        rng = np.random.default_rng(L)
        b = rng.standard_normal((n, n))
        ref = po.Solver(finest_level=L, coarsest_level=L, op=po.OP_STENCIL5, schedule=0)
        ref.set_coefficient(a)
        with pkg.Multigrid(finest_level=L, coarsest_level=L, op=VAR, schedule=0) as mg:
            mg.set_coefficient(a)
            x = mg.bottom_solve(b)                               # MF:63-72
        assert np.array_equal(x, ref.bottom_solve(b))            # same elimination, same order: same bits
        coef = po.stencil_from_nodes(a, L, L)
        assert np.max(np.abs(po.var_residual(x, b, coef))) <= 1e-12 * np.max(np.abs(b))


def test_configurations_the_path_does_not_support_are_refused(pkg):
    for bad in (dict(op=VAR, dtype=2), dict(op=VAR, smoother=1), dict(op=VAR, arith=1), dict(op=VAR, coarsest_level=6),
                dict(op=VAR, n_gpus=2, devices=[0, 0]), dict(restrict_mode=2, dtype=2), dict(restrict_mode=4), dict(op=2)):
        with pytest.raises(pkg.MgxError):
            pkg.Multigrid(**dict(dict(finest_level=8, coarsest_level=4), **bad))
    with pkg.Multigrid(finest_level=6, coarsest_level=4) as mg:
        with pytest.raises(pkg.MgxError, match="POISSON"):
            mg.set_coefficient(np.ones((65, 65)))
    with pkg.Multigrid(finest_level=6, coarsest_level=4, op=VAR) as mg:
        with pytest.raises(pkg.MgxError, match=r"\(N \+ 1\)\^2"):
            mg.set_coefficient(np.ones((64, 64)))
        mg.set_stencil(6, *[np.ones((63, 63))] * 5)
        with pytest.raises(pkg.MgxError, match="level 4 not set"):
            mg.vcycle(6)


def test_seeded_fuzz_of_general_operator_configurations(pkg, po):
    """16 random configurations (fixed seed) of the general-operator path: levels, sweeps, weights, restriction
    modes, schedules, bottom modes, random positive coefficients - histories against the oracle to 1e-10"""
    rng = np.random.default_rng(20261005)
    for case in range(16):
        finest = int(rng.integers(5, 10))
        coarsest = int(rng.integers(2, min(5, finest) + 1))
        cfg = dict(finest_level=finest, coarsest_level=coarsest, mu0=int(rng.integers(0, 2)), mu1=int(rng.integers(0, 5)),
                   mu2=int(rng.integers(0, 5)), omega=float(rng.choice([2.0 / 3.0, 0.8, 0.6])), schedule=int(rng.integers(0, 2)),
                   restrict_mode=int(rng.choice([0, 0, 3, 2, 1])), bottom=int(rng.choice([0, 0, 1])), op=VAR)
        if cfg["mu1"] + cfg["mu2"] == 0:
            cfg["mu2"] = 2
        n = (1 << finest) - 1
        x = np.linspace(0.0, 1.0, n + 2)
        a = np.exp(rng.uniform(-0.7, 0.7) * np.sin(rng.integers(1, 4) * np.pi * x)[None, :] * np.cos(rng.integers(1, 4) * np.pi * x)[:, None])
        a = a * (1.0 + 0.1 * rng.random(a.shape))
        b = po.rhs_sine(finest) if case % 2 else po.rhs_constant(finest)
        u0 = po.fill_uniform((n, n), 500 + case) if cfg["schedule"] == 0 and case % 3 == 0 else None
        with pkg.Multigrid(**cfg) as mg:
            mg.set_coefficient(a)
            mg.set_rhs(b)
            if u0 is not None:
                mg.set_guess(u0)
            st, h = mg.solve(tol=1e-9, max_cycles=6)
            u = mg.get_solution()
        ref = po.Solver(**cfg)
        ref.set_coefficient(a)
        u_ref, h_ref = ref.solve(b, u0, tol=1e-9, max_cycles=6)
        assert hist_close(h, h_ref), (case, cfg, h, h_ref)
        assert np.max(np.abs(u - u_ref)) <= 1e-10 * max(np.max(np.abs(u_ref)), 1e-300), (case, cfg)
