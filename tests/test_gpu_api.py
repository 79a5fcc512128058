"""GPU checks of the rest of the C-ABI surface: device-resident set/get, fills,
profile and timing entry points, error paths, degenerate hierarchies."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_device_resident_set_get_roundtrip(pkg, po):
    import hipmem as hm

    L = pkg.lib()
    level = 8
    N = 1 << level
    rng = np.random.default_rng(1)
    a = rng.uniform(-1, 1, (N - 1, N - 1))
    pitch = L.mgx_level_pitch(level, pkg.DTYPE_F64)
    g = np.zeros((N + 1, pitch))
    g[1:N, 1:N] = a
    t = hm.from_numpy(g)
    with pkg.Multigrid(finest_level=level, coarsest_level=6) as mg:
        mg.set_level_device(level, pkg.VEC_B, t.data_ptr())
        assert np.array_equal(mg.get_level(level, pkg.VEC_B), a)
        out = hm.empty_like(t)
        mg.set_level(level, pkg.VEC_U, 2 * a)
        mg.get_level_device(level, pkg.VEC_U, out.data_ptr())
        hm.synchronize()
        assert np.array_equal(out.cpu().numpy()[1:N, 1:N], 2 * a)
        assert np.all(out.cpu().numpy()[0] == 0) and np.all(out.cpu().numpy()[:, 0] == 0)
        mg.zero_level(level, pkg.VEC_U)
        assert np.all(mg.get_level(level, pkg.VEC_U) == 0)


def test_builtin_right_hand_sides_and_guess(pkg, po):
    with pkg.Multigrid(finest_level=9, coarsest_level=7) as mg:
        mg.fill_rhs(0, 4.0)                                    # PS:283-335: f h^2
        assert np.array_equal(mg.get_level(9, pkg.VEC_B), po.rhs_constant(9, 4.0))
        mg.fill_rhs(1, 0.0)
        b = mg.get_level(9, pkg.VEC_B)
        ref = po.rhs_sine(9)
        assert np.max(np.abs(b - ref)) <= 1e-15 * np.max(np.abs(ref)) * 8
        mg.fill_guess_random(12345)
        u = mg.get_solution()
        assert u.min() >= -1 and u.max() < 1 and abs(u.mean()) < 1e-2 and 0.3 < u.var() < 0.36
        mg.fill_guess_random(12345)
        assert np.array_equal(mg.get_solution(), u)            # deterministic in (seed, index)


def test_profile_and_timing_entry_points(pkg):
    with pkg.Multigrid(finest_level=10, coarsest_level=7, mu1=2, mu2=1, schedule=0, profile=1) as mg:
        mg.fill_rhs(1, 0.0)
        mg.profile_reset()
        st, h = mg.solve(tol=0.0, max_cycles=3)
        p = mg.profile()
        assert p["sweeps"][0] == 3 * 3                          # mu1 + mu2 sweeps per cycle on the finest level
        assert 1 <= p["launches"][0] <= p["sweeps"][0] and p["ms"][0] > 0
        assert p["launches"][4] > 0 and p["ms"][4] > 0          # coarse levels
        ms = mg.time_smoother(4)
        assert ms > 0
        n = (1 << 10) - 1
        assert st.fine_updates == 3 * 3 * n * n and st.history_len == 4 and st.seconds > 0


@pytest.mark.parametrize("smoother,mu1,mu2,finest", [(0, 10, 10, 11), (0, 2, 1, 10), (1, 2, 2, 10), (0, 3, 0, 9)])
def test_profile_two_times_the_fine_passes_around_one_graph_of_the_rest(pkg, smoother, mu1, mu2, finest):
    """cfg.profile = 2 (what bench.py times): the finest level's passes launched one by one between HIP
    events, everything below replayed from one graph between two events - the same bits as the whole-cycle
    graph (profile 0) and as the eager path (profile 1), graphs really cached, phases accounted"""
    kw = dict(finest_level=finest, coarsest_level=6, mu1=mu1, mu2=mu2, smoother=smoother, schedule=0)
    out = {}
    for prof in (0, 1, 2):
        with pkg.Multigrid(profile=prof, **kw) as mg:
            mg.fill_rhs(1, 0.0)
            mg.fill_guess_random(4242)
            if prof:
                mg.profile_reset()
            st, h = mg.solve(tol=0.0, max_cycles=4)
            out[prof] = (mg.get_solution(), h, mg.graphs_cached(), mg.profile() if prof else None, st)
    for prof in (1, 2):
        assert np.array_equal(out[prof][0], out[0][0])
        assert np.allclose(out[prof][1], out[0][1], rtol=1e-13, atol=0)
    assert out[0][2] >= 1 and out[1][2] == -1 and out[2][2] >= 1
    p1, p2 = out[1][3], out[2][3]
    assert p2["sweeps"][0] == p1["sweeps"][0] == 4 * (mu1 + mu2)
    assert p2["launches"][0] == p1["launches"][0]
    assert p2["launches"][4] == 4 and p2["ms"][4] > 0          # the coarse part: one graph launch per cycle
    assert p2["ms"][0] > 0 and out[2][4].fine_updates == out[0][4].fine_updates


def test_error_paths_return_status_not_crash(pkg):
    L = pkg.lib()
    with pkg.Multigrid(finest_level=8, coarsest_level=6, bottom=pkg.BOTTOM_SMOOTH) as mg:
        with pytest.raises(pkg.MgxError, match="level out of range"):
            mg.smooth(9, 1)
        with pytest.raises(pkg.MgxError, match="level out of range"):
            mg.get_level(5, pkg.VEC_U)
        with pytest.raises(pkg.MgxError):
            mg.set_level(8, pkg.VEC_U, np.zeros((10, 10)))      # wrong length
        with pytest.raises(pkg.MgxError, match="SMOOTH"):
            mg.bottom_solve(np.zeros((63, 63)))
        with pytest.raises(pkg.MgxError):
            mg.get_level(8, pkg.VEC_R)                          # residual not computed yet
        assert L.mgx_solve(mg._h, -1.0, 3, None, None, 0) == 1  # negative tolerance
    assert L.mgx_destroy(None) == 0
    assert L.mgx_smooth(None, 3, 1) == 1


def test_single_level_hierarchy_is_a_direct_solve(pkg, po):
    b = po.rhs_sine(7)
    with pkg.Multigrid(finest_level=7, coarsest_level=7, schedule=0) as mg:
        mg.set_rhs(b)
        st, h = mg.solve(tol=1e-12, max_cycles=5)
        u = mg.get_solution()
    assert st.cycles == 1 and st.converged and h[1] <= 1e-12 * h[0]
    ref = po.Solver(finest_level=7, coarsest_level=7, schedule=0)
    u_ref, _ = ref.solve(b, tol=1e-12, max_cycles=5)
    assert np.max(np.abs(u - u_ref)) <= 1e-11 * np.max(np.abs(u_ref))


@pytest.mark.parametrize("cfg", [dict(smoother=1, dtype=0, schedule=1, mu0=1, mu1=1, mu2=1),
                                 dict(smoother=1, dtype=2, schedule=1, mu0=0, mu1=2, mu2=1),
                                 dict(smoother=0, dtype=2, schedule=0, mu1=3, mu2=3)])
def test_other_smoother_precision_schedule_combinations(pkg, po, cfg):
    full = dict(finest_level=9, coarsest_level=6, **cfg)
    b = po.rhs_constant(9)
    with pkg.Multigrid(**full) as mg:
        mg.set_rhs(b)
        st, h = mg.solve(tol=1e-8, max_cycles=12)
        u = mg.get_solution()
    u_ref, h_ref = po.Solver(**full).solve(b, None, tol=1e-8, max_cycles=12)
    assert len(h) == len(h_ref)
    tol = 2e-3 if cfg["dtype"] == 0 else 1e-3
    keep = h_ref > 1e-4 * h_ref[0] if cfg["dtype"] == 0 else np.ones(len(h_ref), bool)
    assert np.all(np.abs(h[keep] - h_ref[keep]) <= tol * h_ref[keep] + 1e-13 * h_ref[0]), (h, h_ref)
    assert np.max(np.abs(u - u_ref)) <= (1e-4 if cfg["dtype"] == 0 else 1e-8) * np.max(np.abs(u_ref))


def test_nonzero_dirichlet_data_known_answer(pkg, po):
    """u = x^2 - y^2 is harmonic and the 5-point Laplacian is exact for quadratics, so with
    f = 0 and g = u on the boundary the discrete solution IS u at the nodes (to rounding)."""
    level = 9
    N = 1 << level
    x = np.arange(N + 1) / N
    U = x[None, :] ** 2 - x[:, None] ** 2             # rows = y, columns = x
    with pkg.Multigrid(finest_level=level, coarsest_level=6, mu1=2, mu2=2, schedule=0, smoother=1) as mg:
        mg.set_rhs_dirichlet(np.zeros((N - 1, N - 1)), U[0, :], U[N, :], U[1:N, 0], U[1:N, N])
        st, h = mg.solve(tol=1e-13, max_cycles=30)
        u = mg.get_solution()
    assert st.converged
    assert np.max(np.abs(u - U[1:N, 1:N])) < 1e-11
    # and the folding itself against a numpy statement
    b = np.zeros((N - 1, N - 1))
    b[0, :] += U[0, 1:N]; b[-1, :] += U[N, 1:N]; b[:, 0] += U[1:N, 0]; b[:, -1] += U[1:N, N]
    with pkg.Multigrid(finest_level=level, coarsest_level=6) as mg:
        mg.set_rhs_dirichlet(np.zeros((N - 1, N - 1)), U[0, :], U[N, :], U[1:N, 0], U[1:N, N])
        assert np.array_equal(mg.get_level(level, pkg.VEC_B), b)
        with pytest.raises(pkg.MgxError, match="ring"):
            pkg.lib()  # keep lib loaded
            mg._chk(pkg.lib().mgx_set_rhs_dirichlet(mg._h, b.ctypes.data, b.size, b.ctypes.data, 7), "mgx_set_rhs_dirichlet")


@pytest.mark.parametrize("graph", ["1", "0"])
@pytest.mark.parametrize("cfg", [dict(finest_level=9, coarsest_level=6, mu1=3, mu2=2), dict(finest_level=11, coarsest_level=7, mu1=10, mu2=10),
                                 dict(finest_level=8, coarsest_level=5, mu1=2, mu2=1, smoother=1), dict(finest_level=7, coarsest_level=7),
                                 dict(finest_level=9, coarsest_level=6, mu1=0, mu2=2)])
def test_vcycle_zero_equals_zero_guess_then_vcycle(pkg, po, monkeypatch, cfg, graph):
    """mgx_vcycle_zero (implicit zero guess, hipGraph replay) against zero_level + mgx_vcycle, called
    several times on changing right-hand sides (what the multi-GPU driver's coarse solver does)"""
    monkeypatch.setenv("MGX_GRAPH", graph)
    L = cfg["finest_level"]
    n = (1 << L) - 1
    rng = np.random.default_rng(5)
    with pkg.Multigrid(**cfg) as a, pkg.Multigrid(**cfg) as b:
        for it in range(4):
            f = rng.uniform(-1, 1, (n, n))
            a.set_rhs(f)
            a.set_guess(rng.uniform(-1, 1, (n, n)))      # stale data the call must ignore
            a.vcycle_zero()
            b.set_rhs(f)
            b.zero_level(L, pkg.VEC_U)
            b.vcycle(L)
            assert np.array_equal(a.get_solution(), b.get_solution()), (cfg, it)
        if graph == "1":
            assert 1 <= a.graphs_cached() <= 2


def test_this_process_runs_one_rocm_stack_the_one_libmgx_was_built_against(pkg):
    """libmgx.so is built by /opt/rocm's hipcc with RUNPATH /opt/rocm/lib; the torch wheel bundles another copy of
    libamdhip64 / libhsa-runtime64 / librccl with the same SONAMEs, and whichever is loaded first serves the whole
    process.  The GPU tests (the records the driver keeps) must exercise the library on ITS stack: exactly one copy
    of each runtime library mapped, from /opt/rocm, and no torch in sys.modules."""
    import sys

    with pkg.Multigrid(finest_level=6, coarsest_level=5) as mg:
        mg.fill_rhs(1, 0.0)
    libs = pkg.runtime_libs()
    for name in ("libamdhip64", "libhsa-runtime64", "librccl"):
        paths = [p for p in libs if name in p]
        assert len(paths) == 1 and paths[0].startswith("/opt/rocm"), libs
    assert "torch" not in sys.modules
