"""The oracle's general per-level operator path (oracle/mg_oracle_var.inc; MF = Multigrid_functions.cpp:
ProblemVar MF:16-41, jacobirelaxation MF:75-96, restriction2D MF:122-130, residual MF:150-153, direct bottom
solve MF:63-72) - SURVEY §8(f)4.

Pinned by reference-computed data where the reference can be run: the COO triplets its own assembly
(globalstiffenssmatrix PS:200-281, fixture tests/golden/ref_ps.npz) emits are turned into CSR matrices in
the reference's data layout (MF:33-41) and fed to the oracle's CSR kernels; the five-coefficient-array
kernels the device mirrors must reproduce those results bit for bit on the constant-coefficient operator.
The variable-coefficient case has no reference data (MF's front-end was never written): parity unpinned,
checked against numpy statements, a manufactured solution and convergence factors."""
import os

import numpy as np
import pytest

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_ps.npz")


def ref_csr(nodes):
    """A (SPD sign, D1) of the reference's own assembly as CSR: duplicates merged, columns sorted - what a correct
    coo_to_csr (D2, D12) would emit.  Returns (indptr, indices, values, n)."""
    ref = np.load(GOLD)
    n = nodes - 2
    N = n * n
    A = np.zeros((N, N))
    for nm in ("lu", "d"):
        np.add.at(A, (ref[f"coo{nodes}_rows_{nm}"], ref[f"coo{nodes}_cols_{nm}"]), ref[f"coo{nodes}_vals_{nm}"].astype(np.float64))
    A = -A                                                   # D1: the assembled matrix is minus the SPD operator
    indptr, indices, values = [0], [], []
    for r in range(N):
        cols = np.nonzero(A[r])[0]
        indices += cols.tolist()
        values += A[r, cols].tolist()
        indptr.append(len(indices))
    return np.array(indptr, np.int32), np.array(indices, np.int32), np.array(values), n


def jacobi_csr_of(indptr, indices, values, omega, dt):
    """{D_inv, R_omega} (MF:28-32) of a CSR operator, in the operation order of orc_var_build_jacobi"""
    values = values.astype(dt)
    N = len(indptr) - 1
    om = dt(omega)
    dinv = np.empty(N, dt)
    r_values = np.empty_like(values)
    for r in range(N):
        sl = slice(indptr[r], indptr[r + 1])
        d = dt(1) / values[sl][indices[sl] == r][0]
        dinv[r] = d
        r_values[sl] = np.where(indices[sl] == r, dt(1.0 - float(om)), -(om * (d * values[sl])))
    return dinv, r_values


def stencil_arrays(n, dt=np.float64):
    c = np.full((n, n), 4.0, dt)
    o = np.full((n, n), -1.0, dt)
    return c, o, o.copy(), o.copy(), o.copy()


@pytest.mark.parametrize("nodes", [9, 17])
@pytest.mark.parametrize("dt", [np.float64, np.float32])
def test_stencil_array_kernels_equal_the_csr_kernels_on_the_reference_matrix(po, nodes, dt):
    indptr, indices, values, n = ref_csr(nodes)
    rng = np.random.default_rng(nodes)
    v = rng.standard_normal((n, n)).astype(dt)
    b = rng.standard_normal((n, n)).astype(dt)
    coef = stencil_arrays(n, dt)
    # A v: the reference's matrix through the reference's data layout == the five-array form == the Poisson oracle
    av = po.csr_gemv(indptr, indices, values.astype(dt), v.ravel()).reshape(n, n)
    assert np.array_equal(b - av, po.var_residual(v, b, coef))
    assert np.max(np.abs((b - av) - po.residual(v, b))) <= 8 * np.finfo(dt).eps * np.max(np.abs(av))
    # MF:75-96 on the CSR {D_inv, R_omega} of that matrix == the five-array sweeps, bit for bit, several sweeps
    for omega in (2.0 / 3.0, 0.8):
        dinv, r_values = jacobi_csr_of(indptr, indices, values, omega, dt)
        want = po.csr_jacobi(v.ravel(), b.ravel(), 4, omega, indptr, indices, r_values, dinv).reshape(n, n)
        jac = po.var_build_jacobi(*coef, omega=omega)
        assert np.array_equal(jac[0].ravel(), dinv)
        got = po.var_jacobi(v, b, 4, omega, jac)
        assert np.array_equal(got, want)
        # and MF's form is the reference's first-draft Jacobi (PS:125-147) up to rounding: same iteration
        assert np.max(np.abs(got - po.jacobi(v, b, 4, omega))) <= 32 * np.finfo(dt).eps * max(1.0, np.max(np.abs(got)))


def test_csr_gemv_is_the_in_order_row_sum_with_alpha(po):
    indptr, indices, values, n = ref_csr(9)
    x = np.arange(1.0, n * n + 1)
    y = po.csr_gemv(indptr, indices, values, x, alpha=0.5)
    A = np.zeros((n * n, n * n))
    for r in range(n * n):
        A[r, indices[indptr[r]:indptr[r + 1]]] = values[indptr[r]:indptr[r + 1]]
    assert np.array_equal(y, 0.5 * (A @ x))               # small integers: exact


def test_injection_takes_the_coincident_node(po):
    rng = np.random.default_rng(3)
    fine = rng.standard_normal((15, 15))
    c = po.restrict_inject(fine)                            # MF:126-128
    assert c.shape == (7, 7) and np.array_equal(c, fine[1::2, 1::2])
    assert np.array_equal(po.restrict_inject(fine, 4.0), 4.0 * fine[1::2, 1::2])
    # injection is exact on what bilinear interpolation produces: inject(P e) = e
    e = rng.standard_normal((7, 7))
    assert np.array_equal(po.restrict_inject(po.prolong(e)), e)


def coefficient(L, kind):
    x = np.linspace(0.0, 1.0, (1 << L) + 1)
    if kind == "one":
        return np.ones((len(x), len(x)))
    if kind == "smooth":
        return 1.0 + 0.8 * np.sin(3 * np.pi * x)[None, :] * np.cos(2 * np.pi * x)[:, None]
    rng = np.random.default_rng(11)                          # "jump": piecewise constant, contrast 4 (re-discretised geometric multigrid is not robust for large jumps)
    a = np.ones((len(x), len(x)))
    a[(x[:, None] > 0.3) & (x[:, None] < 0.7) & (x[None, :] > 0.25) & (x[None, :] < 0.6)] = 4.0
    return a * (1.0 + 0.05 * rng.random(a.shape))


def test_stencil_from_nodes_is_symmetric_and_conservative(po):
    L = 5
    a = coefficient(L, "smooth")
    c, n, s, w, e = po.stencil_from_nodes(a, L, L)
    assert np.all(c > 0) and np.all(n < 0) and np.all(s < 0) and np.all(w < 0) and np.all(e < 0)
    assert np.array_equal(e[:, :-1], w[:, 1:]) and np.array_equal(s[:-1, :], n[1:, :])        # symmetric operator
    assert np.max(np.abs(c + n + s + w + e)) <= 4 * np.finfo(float).eps * np.max(c)            # zero row sums
    # coarse levels sample the same coefficient at their own nodes
    c2, *_ = po.stencil_from_nodes(a, L - 1, L)
    c2b, *_ = po.stencil_from_nodes(a[::2, ::2], L - 1, L - 1)
    assert np.array_equal(c2, c2b)


def test_constant_coefficient_hierarchy_follows_the_poisson_path(po):
    L = 7
    b = po.rhs_sine(L)
    u0 = po.fill_uniform(b.shape, 5)
    kw = dict(finest_level=L, coarsest_level=4, mu1=2, mu2=1, schedule=0)
    sv = po.Solver(op=po.OP_STENCIL5, **kw)
    sv.set_coefficient(coefficient(L, "one"))
    u1, h1 = sv.solve(b, u0, tol=1e-9, max_cycles=40)
    u0_, h0 = po.Solver(**kw).solve(b, u0, tol=1e-9, max_cycles=40)
    assert len(h1) == len(h0) and np.all(np.abs(h1 - h0) <= 1e-10 * h0 + 1e-13 * h0[0])
    assert np.max(np.abs(u1 - u0_)) <= 1e-11 * np.max(np.abs(u0_))


@pytest.mark.parametrize("kind,mode,rho_max", [("smooth", 0, 0.3), ("smooth", 3, 0.4), ("jump", 0, 0.6), ("smooth", 2, 0.995)])
def test_variable_coefficient_cycles_converge(po, kind, mode, rho_max):
    """full weighting (PS:531-546, consistent weight) and 4 x injection converge at multigrid rates on smooth
    coefficients; MF's injection as written (weight 1) leaves the coarse correction a quarter of its size -
    the same scaling defect as D4 - and crawls"""
    L = 7
    a = coefficient(L, kind)
    coef = po.stencil_from_nodes(a, L, L)
    x = np.linspace(0, 1, (1 << L) + 1)[1:-1]
    ut = np.sin(np.pi * x)[None, :] * np.sin(2 * np.pi * x)[:, None]
    b = -po.var_residual(ut, np.zeros_like(ut), coef)                    # A ut
    sv = po.Solver(finest_level=L, coarsest_level=3, mu1=2, mu2=2, schedule=0, op=po.OP_STENCIL5, restrict_mode=mode)
    sv.set_coefficient(a)
    u, h = sv.solve(b, None, tol=1e-10, max_cycles=25 if mode != 2 else 12)
    rho = h[1:] / h[:-1]
    assert np.all(rho[1:] < rho_max), rho
    if mode != 2:
        assert h[-1] <= 1e-10 * h[0] and np.max(np.abs(u - ut)) <= 1e-8
    else:
        assert rho[-1] > 0.9


def test_dense_bottom_solve_is_exact(po):
    L = 4
    a = coefficient(L, "smooth")
    sv = po.Solver(finest_level=L, coarsest_level=L, op=po.OP_STENCIL5, schedule=0)
    sv.set_coefficient(a)
    n = (1 << L) - 1
    rng = np.random.default_rng(2)
    b = rng.standard_normal((n, n))
    x = sv.bottom_solve(b)
    coef = po.stencil_from_nodes(a, L, L)
    assert np.max(np.abs(po.var_residual(x, b, coef))) <= 1e-12 * np.max(np.abs(b))
    u, h = sv.solve(b, None, tol=1e-12, max_cycles=3)
    assert len(h) == 2 and h[1] <= 1e-12 * h[0]


def test_fmg_on_general_operators(po):
    L = 6
    a = coefficient(L, "smooth")
    b = po.rhs_sine(L)
    for mode, first in ((0, 0.05), (3, 0.2)):
        sv = po.Solver(finest_level=L, coarsest_level=3, mu0=0, mu1=2, mu2=2, schedule=1, op=po.OP_STENCIL5, restrict_mode=mode)
        sv.set_coefficient(a)
        u, h = sv.solve(b, None, tol=1e-9, max_cycles=20)
        assert h[-1] <= 1e-9 * h[0] and h[1] < first * h[0]   # the FMG pass alone removes most of the residual
