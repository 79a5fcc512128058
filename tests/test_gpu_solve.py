"""GPU parity of the schedules (V-cycle, FMG, mixed precision) and whole solves:
residual-norm histories against the oracle within 1e-10 relative per cycle
(BASELINE north_star), against the committed fixtures, and size-independent
properties at BASELINE's full sizes."""
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
HIST_TOL = 1e-10       # relative, per cycle (north_star)
# Below ~1e-13 ||r0|| a double-precision residual is rounding noise: evaluating
# b - A u costs eps * (|b| + 8|u|) per entry, i.e. ~6e-14 in the 2-norm at
# n = 255 and more on larger grids, and the device's FMA contraction moves u by
# an ulp.  Histories are therefore compared to 1e-10 relative plus that floor.
HIST_FLOOR = 1e-13


def hist_close(h, ref, rtol=HIST_TOL, floor_abs=0.0):
    h, ref = np.asarray(h), np.asarray(ref)
    return len(h) == len(ref) and bool(np.all(np.abs(h - ref) <= rtol * ref + max(HIST_FLOOR * ref[0], floor_abs)))


def oracle_cfg(po, cfg):
    """The oracle configuration that mirrors a device configuration.  For float hierarchies
    (dtype F32 / MIXED) with an exact bottom solve the oracle uses its sine-transform bottom mode
    (ORC_BOTTOM_DST: the device's direct method in the device's operation order), so that both
    sides round THE SAME fp64 bottom solution to float; the float paths are then held to the same
    1e-10 as double instead of the 1e-3 the Cholesky-vs-sine-transform last bit used to need.
    Double hierarchies keep the Cholesky solve: an independent exact method, well inside 1e-10."""
    c = dict(cfg)
    if c.get("dtype", 1) != 1 and c.get("bottom", 0) == 0:
        c["bottom"] = po.BOTTOM_DST
    return c


def problem(po, L, rhs):
    if rhs == "constant":
        return po.rhs_constant(L), None
    return po.rhs_sine(L), po.fill_uniform(((1 << L) - 1,) * 2, 12345)


def run_gpu(pkg, cfg, b, u0, tol=1e-8, max_cycles=20):
    with pkg.Multigrid(**cfg) as mg:
        mg.set_rhs(b)
        if u0 is not None:
            mg.set_guess(u0)
        st, h = mg.solve(tol=tol, max_cycles=max_cycles)
        u = mg.get_solution()
    return st, h, u


def test_histories_match_committed_fixtures(pkg, po):
    gold = json.load(open(os.path.join(GOLD, "histories.json")))
    for key, rec in gold.items():
        cfg = rec["cfg"]
        b, u0 = problem(po, cfg["finest_level"], key.split("/")[1])
        st, h, u = run_gpu(pkg, cfg, b, u0)
        ref = np.array(rec["history"])
        # float / mixed fixtures were written with the oracle's sine-transform bottom mode
        # (oracle_cfg above): every history, float ones included, is held to 1e-10
        assert len(h) == len(ref), (key, len(h), len(ref))
        assert hist_close(h, ref), (key, h, ref)
        n = u.shape[0]
        assert abs(u[n // 2, n // 2] - rec["u_centre"]) <= 1e-9 * max(1.0, abs(rec["u_absmax"])), key


@pytest.mark.parametrize(
    "cfg",
    [
        # BASELINE config 1 (CPU reference case) in the reference's own parameters
        dict(finest_level=8, coarsest_level=6, mu1=10, mu2=10, schedule=0),
        dict(finest_level=8, coarsest_level=6, mu1=2, mu2=1, schedule=0),
        # config 2 shape: 6-level Jacobi V-cycle (oracle-sized: 1024^2)
        dict(finest_level=10, coarsest_level=5, mu1=2, mu2=1, schedule=0),
        # reference hierarchy 10..7 (PS:17-18) with its FMG schedule
        dict(finest_level=10, coarsest_level=7, mu0=1, mu1=2, mu2=2, schedule=1),
        # config 3 shape: RB-GS
        dict(finest_level=10, coarsest_level=5, mu1=2, mu2=1, schedule=0, smoother=1),
        dict(finest_level=9, coarsest_level=7, mu1=1, mu2=1, schedule=0, smoother=1),
    ],
)
@pytest.mark.parametrize("rhs", ["constant", "sine_random_guess"])
def test_f64_history_matches_oracle(pkg, po, cfg, rhs):
    b, u0 = problem(po, cfg["finest_level"], rhs)
    st, h, u = run_gpu(pkg, cfg, b, u0, max_cycles=25)
    u_ref, h_ref = po.Solver(**cfg).solve(b, u0, tol=1e-8, max_cycles=25)
    assert hist_close(h, h_ref), (h, h_ref)
    assert np.max(np.abs(u - u_ref)) <= 1e-11 * np.max(np.abs(u_ref))
    assert st.converged == 1 and st.cycles == len(h) - 1


def test_single_vcycle_and_fmg_entry_points(pkg, po):
    cfg = dict(finest_level=9, coarsest_level=6, mu0=0, mu1=3, mu2=2)
    b = po.rhs_sine(9)
    u0 = po.fill_uniform(b.shape, 7)
    ref = po.Solver(**cfg)
    with pkg.Multigrid(**cfg) as mg:
        v = mg.vcyclemultigrid(9, u0, b)                    # PS:575
        assert np.max(np.abs(v - ref.vcycle(9, u0, b))) <= 1e-12 * np.max(np.abs(v))
        # a V-cycle from an intermediate level (PS:617 recursion entry)
        n7 = (1 << 7) - 1
        v7 = mg.vcyclemultigrid(7, u0[:n7, :n7], b[:n7, :n7])
        assert np.max(np.abs(v7 - ref.vcycle(7, u0[:n7, :n7].copy(), b[:n7, :n7].copy()))) <= 1e-12 * np.max(np.abs(v7))
        w = mg.fullmultigrid(b)                             # PS:629
        assert np.max(np.abs(w - ref.fmg(9, b))) <= 1e-12 * np.max(np.abs(w))


def test_reference_literal_bottom_and_weights(pkg, po):
    """the as-written choices (D4 weights, D8 smoothed bottom) remain selectable
    and match the oracle too"""
    b = po.rhs_constant(8)
    for extra in (dict(bottom=1), dict(restrict_mode=1), dict(bottom=1, restrict_mode=1)):
        cfg = dict(finest_level=8, coarsest_level=6, mu1=10, mu2=10, schedule=0, **extra)
        st, h, u = run_gpu(pkg, cfg, b, None, max_cycles=8)
        _, h_ref = po.Solver(**cfg).solve(b, None, tol=1e-8, max_cycles=8)
        assert hist_close(h, h_ref), (h, h_ref)


def test_mixed_precision_tracks_f64(pkg, po):
    # BASELINE config 5 shape at oracle size
    cfg = dict(finest_level=10, coarsest_level=7, mu0=0, mu1=2, mu2=1, schedule=1, dtype=2)
    b = po.rhs_constant(10)
    st, h, u = run_gpu(pkg, cfg, b, None, max_cycles=25)
    u_ref, h_ref = po.Solver(**oracle_cfg(po, cfg)).solve(b, None, tol=1e-8, max_cycles=25)
    # the inner cycle is fp32 and performs the oracle's operations in the oracle's order, the
    # bottom solve included (ORC_BOTTOM_DST): same tolerance as double
    assert hist_close(h, h_ref), (h, h_ref)
    assert h[-1] <= 1e-8 * h[0]
    assert np.max(np.abs(u - u_ref)) <= 1e-12 * np.max(np.abs(u_ref))
    # and the same iteration counts as full double (SURVEY §6.2 last row)
    cfg64 = dict(cfg, dtype=1)
    _, h64 = po.Solver(**cfg64).solve(b, None, tol=1e-8, max_cycles=25)
    assert len(h) == len(h64)


def test_pure_f32_stalls_like_the_reference_precision(pkg, po):
    cfg = dict(finest_level=10, coarsest_level=7, mu1=2, mu2=1, schedule=0, dtype=0)
    st, h, u = run_gpu(pkg, cfg, po.rhs_constant(10), None, max_cycles=20)
    assert st.converged == 0 and 1e-3 < h[-1] / h[0] < 5e-2      # D11


# ---- BASELINE's full sizes: size-independent properties ------------------------------
def _check_known_answer(u, tol):
    n = u.shape[0]
    assert abs(u[n // 2, n // 2] - 0.2946854) < tol
    s = u[:: max(1, n // 512), :: max(1, n // 512)]
    assert np.max(np.abs(s - s.T)) < 1e-12
    assert np.max(np.abs(u[:257, :257] - u[::-1, ::-1][:257, :257][::1, ::1])) < 1e-12


def test_config2_4096_six_level_jacobi(pkg):
    cfg = dict(finest_level=12, coarsest_level=7, mu1=2, mu2=1, schedule=0)
    with pkg.Multigrid(**cfg) as mg:
        mg.fill_rhs(0, 4.0)
        st, h = mg.solve(tol=1e-8, max_cycles=30)
        u = mg.get_solution()
    assert st.converged and 14 <= st.cycles <= 17          # h-independent count (15-16)
    rho = (h[1:] / h[:-1])[2:]
    assert np.all(rho < 0.4)
    _check_known_answer(u, 1e-6)


def test_config3_8192_rbgs(pkg):
    cfg = dict(finest_level=13, coarsest_level=7, mu1=2, mu2=1, schedule=0, smoother=1)
    with pkg.Multigrid(**cfg) as mg:
        mg.fill_rhs(0, 4.0)
        st, h = mg.solve(tol=1e-8, max_cycles=20)
        u = mg.get_solution()
    assert st.converged and 6 <= st.cycles <= 8            # 7
    _check_known_answer(u, 1e-6)


def test_config5_8192_fmg_mixed(pkg):
    cfg = dict(finest_level=13, coarsest_level=7, mu0=0, mu1=2, mu2=1, schedule=1, dtype=2)
    with pkg.Multigrid(**cfg) as mg:
        mg.fill_rhs(0, 4.0)
        st, h = mg.solve(tol=1e-8, max_cycles=30)
        u = mg.get_solution()
    # a float FMG pass leaves the residual on the float rounding floor (D11: about
    # 0.5 ||b|| at n = 8191), but its error is already small; the double defect
    # correction then converges at the V-cycle rate
    assert st.converged and st.cycles <= 17
    assert np.all((h[2:] / h[1:-1])[1:] < 0.4)
    _check_known_answer(u, 1e-6)


def test_config1_parameters_at_8192_jacobi_v1010(pkg):
    # the reference's own V(10,10), omega = 2/3 on the metric's grid
    cfg = dict(finest_level=13, coarsest_level=7, mu1=10, mu2=10, schedule=0)
    with pkg.Multigrid(**cfg) as mg:
        mg.fill_rhs(0, 4.0)
        st, h = mg.solve(tol=1e-8, max_cycles=12)
    assert st.converged and 5 <= st.cycles <= 7            # 6 at every size
    n = float((1 << 13) - 1)
    assert st.fine_updates == 20.0 * st.cycles * n * n


@pytest.mark.parametrize("tiles", ["tiles", "marching"])
@pytest.mark.parametrize("dtype", [1, 0])
@pytest.mark.parametrize("mu1,mu2", [(10, 10), (2, 1), (1, 1), (5, 3), (3, 0), (6, 8), (7, 7), (9, 8), (8, 1), (12, 11)])
def test_folded_cycle_passes_are_bit_identical(pkg, po, monkeypatch, dtype, mu1, mu2, tiles):
    """k_jacobi_cycle / k_tile_smooth (correction on load, residual+restriction and ||r||^2
    appended to the smoother passes) must give the bits of the stand-alone kernels, and both
    must match the oracle.  Level 11 always marches; 10..8 use LDS tiles or march."""
    if tiles == "marching":
        monkeypatch.setenv("MGX_TILE_MAX_N", "0")
    else:
        monkeypatch.delenv("MGX_TILE_MAX_N", raising=False)
    cfg = dict(finest_level=11, coarsest_level=8, mu1=mu1, mu2=mu2, schedule=0, dtype=dtype)
    b = po.rhs_sine(11)
    u0 = po.fill_uniform(b.shape, 99)
    out = {}
    for fold in ("0", "1"):
        monkeypatch.setenv("MGX_FOLD", fold)
        st, h, u = run_gpu(pkg, cfg, b, u0, tol=0.0, max_cycles=3)
        out[fold] = (h, u)
    assert np.array_equal(out["0"][1], out["1"][1])
    assert np.allclose(out["0"][0], out["1"][0], rtol=1e-13, atol=0)
    if dtype == 1:
        _, h_ref = po.Solver(**cfg).solve(b, u0, tol=0.0, max_cycles=3)
        assert hist_close(out["1"][0], h_ref), (out["1"][0], h_ref)


@pytest.mark.parametrize("smoother", [0, 1])
def test_tuning_knobs_never_change_a_bit(pkg, po, monkeypatch, smoother):
    """MGX_* environment knobs select kernels (marching vs row kernel, fused depth, folding,
    chunk height, fusion threshold); every combination must give the same bits"""
    cfg = dict(finest_level=10, coarsest_level=7, mu1=4, mu2=3, schedule=0, smoother=smoother)
    b = po.rhs_sine(10)
    u0 = po.fill_uniform(b.shape, 5)
    march = [{}, {"MGX_FUSE": "1"}, {"MGX_FUSE": "1", "MGX_ROWS": "8"}, {"MGX_FOLD": "0"},
             {"MGX_FOLD_KMAX": "10", "MGX_FUSE_ROWS": "16"}, {"MGX_FUSE_MIN_N": "1024"}, {"MGX_FUSE": "2", "MGX_FUSE_MIN_N": "128"},
             {"MGX_ZERO_IN": "0"}, {"MGX_ZERO_IN": "0", "MGX_FOLD": "0"}]
    knobs = [{}] + [dict(kn, MGX_TILE_MAX_N="0") for kn in march]
    knobs += [{"MGX_TILE_K": "2"}, {"MGX_TILE_K": "5", "MGX_FOLD": "0"}, {"MGX_TILE_MAX_N": "256"},
              {"MGX_ZERO_IN": "0"}, {"MGX_TILE_MAX_N": "512", "MGX_TILE_K": "3", "MGX_ZERO_IN": "0"}]
    # explicit pass plans for the folded blocks, and deeper passes where no residual stage follows
    knobs += [{"MGX_TILE_MAX_N": "0", "MGX_PLAN_MIN_N": "256", "MGX_PLAN_PRE": "3,1", "MGX_PLAN_POST": "1,2"},
              {"MGX_TILE_MAX_N": "0", "MGX_PLAN_MIN_N": "256", "MGX_PLAN_PRE": "1,1,2", "MGX_PLAN_POST": "2,1"},
              {"MGX_TILE_MAX_N": "0", "MGX_PLAN_MIN_N": "256", "MGX_PLAN_PRE": "4", "MGX_PLAN_POST": "3"},
              {"MGX_TILE_MAX_N": "0", "MGX_FOLD_KMAX_NOPOST": "10", "MGX_FOLD_KMAX": "2"},
              {"MGX_TILE_MAX_N": "0", "MGX_FOLD_KMAX_NOPOST": "5"}, {"MGX_TILE_MAX_N": "0", "MGX_FOLD_KMAX_NOPOST": "2"}]
    ref = None
    for kn in knobs:
        for k in ("MGX_FUSE", "MGX_ROWS", "MGX_FOLD", "MGX_FOLD_KMAX", "MGX_FUSE_ROWS", "MGX_FUSE_MIN_N", "MGX_ZERO_IN",
                  "MGX_TILE_MAX_N", "MGX_TILE_K", "MGX_PLAN_MIN_N", "MGX_PLAN_PRE", "MGX_PLAN_POST", "MGX_FOLD_KMAX_NOPOST"):
            monkeypatch.delenv(k, raising=False)
        for k, v in kn.items():
            monkeypatch.setenv(k, v)
        st, h, u = run_gpu(pkg, cfg, b, u0, tol=0.0, max_cycles=2)
        if ref is None:
            ref = u
            _, h_orc = po.Solver(**cfg).solve(b, u0, tol=0.0, max_cycles=2)
            assert hist_close(h, h_orc)
        assert np.array_equal(u, ref), kn


def test_seeded_fuzz_of_configurations_against_the_oracle(pkg, po, monkeypatch):
    """40 random configurations (fixed seed): levels 4..10, 0..6 sweeps, both smoothers, f64 and
    mixed, V and FMG, both restriction weights and bottom modes.  Every history must match the
    oracle's to 1e-10 relative, double and mixed alike (oracle_cfg: same bottom method for the
    float inner cycle); the absolute rounding floor of an exact solve applies to one-level
    hierarchies only."""
    rng = np.random.default_rng(20261004)
    for case in range(40):
        finest = int(rng.integers(6, 11))
        coarsest = int(rng.integers(max(2, finest - 5), finest + 1))
        coarsest = min(coarsest, 8)
        cfg = dict(finest_level=finest, coarsest_level=coarsest,
                   mu0=int(rng.integers(0, 2)), mu1=int(rng.integers(0, 7)), mu2=int(rng.integers(0, 7)),
                   omega=float(rng.choice([2.0 / 3.0, 0.8, 0.5])), smoother=int(rng.integers(0, 2)),
                   dtype=int(rng.choice([1, 1, 2])), schedule=int(rng.integers(0, 2)),
                   restrict_mode=int(rng.choice([0, 0, 1])), bottom=int(rng.choice([0, 0, 1])))
        if cfg["mu1"] + cfg["mu2"] == 0:
            cfg["mu1"] = 1
        n = (1 << finest) - 1
        b = po.rhs_sine(finest) if case % 2 else po.rhs_constant(finest)
        u0 = po.fill_uniform((n, n), 1000 + case) if case % 3 == 0 and cfg["schedule"] == 0 else None
        # LDS tiles on every level (the default) or marching kernels, alternating in pairs
        if (case // 2) % 2:
            monkeypatch.setenv("MGX_TILE_MAX_N", "0")
        else:
            monkeypatch.delenv("MGX_TILE_MAX_N", raising=False)
        st, h, u = run_gpu(pkg, cfg, b, u0, tol=1e-9, max_cycles=6)
        u_ref, h_ref = po.Solver(**oracle_cfg(po, cfg)).solve(b, u0, tol=1e-9, max_cycles=6)
        assert len(h) == len(h_ref), (case, cfg, h, h_ref)
        scale = max(np.max(np.abs(u_ref)), 1e-300)
        # rounding floor of a double residual: eps * (|b| + 8|u|) per entry, n entries per row and
        # column in the 2-norm; it is what is left after an exact solve, i.e. on a one-level
        # hierarchy, where the two exact methods (Cholesky, sine transform) leave different noise
        one_level = (finest == coarsest and cfg["bottom"] == 0)
        floor = 32 * np.finfo(np.float64).eps * scale * n if one_level else 0.0
        assert hist_close(h, h_ref, HIST_TOL, floor), (case, cfg, h, h_ref)
        assert np.max(np.abs(u - u_ref)) <= 1e-10 * scale, (case, cfg)


GRAPH_CASES = [
    # the reference's own hierarchy, tiles on every level: an even number of passes per level
    (dict(finest_level=10, coarsest_level=7, mu1=10, mu2=10, schedule=0), {}),
    # marching kernels with explicit plans: three passes down, two up -> the buffers alternate
    (dict(finest_level=10, coarsest_level=6, mu1=4, mu2=3, schedule=0),
     {"MGX_TILE_MAX_N": "0", "MGX_PLAN_MIN_N": "256", "MGX_PLAN_PRE": "1,1,2", "MGX_PLAN_POST": "2,1"}),
    (dict(finest_level=9, coarsest_level=5, mu1=3, mu2=2, schedule=0, smoother=1), {}),
    (dict(finest_level=11, coarsest_level=8, mu1=2, mu2=1, schedule=0, dtype=0), {}),
    (dict(finest_level=9, coarsest_level=6, mu1=1, mu2=1, schedule=0, bottom=1), {"MGX_FOLD": "0"}),
    (dict(finest_level=9, coarsest_level=6, mu0=0, mu1=2, mu2=2, schedule=1), {}),
]


@pytest.mark.parametrize("case", range(len(GRAPH_CASES)))
def test_graph_replay_of_the_cycle_is_bit_identical(pkg, po, monkeypatch, case):
    """mgx_solve replays "V-cycle + residual norm" from a hipGraph when profiling is off.  The
    graph is keyed by the u/tmp buffer assignment of every level, so: several solves on one
    handle, a cycle with an odd number of passes (assignment alternates), an API call between
    solves that flips a buffer, FMG followed by V-cycles - all must give exactly the bits and
    the history of the direct path (MGX_GRAPH=0)."""
    cfg, env = GRAPH_CASES[case]
    L = cfg["finest_level"]
    n = (1 << L) - 1
    dt = np.float32 if cfg.get("dtype", 1) == 0 else np.float64
    b = po.rhs_sine(L).astype(dt)
    u0 = po.fill_uniform((n, n), 77).astype(dt)
    u1 = po.fill_uniform((n, n), 78).astype(dt)
    for k in ("MGX_TILE_MAX_N", "MGX_PLAN_MIN_N", "MGX_PLAN_PRE", "MGX_PLAN_POST", "MGX_FOLD", "MGX_GRAPH"):
        monkeypatch.delenv(k, raising=False)
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    out = {}
    for graph in ("0", "1"):
        monkeypatch.setenv("MGX_GRAPH", graph)
        res = []
        with pkg.Multigrid(**cfg) as mg:
            mg.set_rhs(b)
            mg.set_guess(u0)
            st, h = mg.solve(tol=0.0, max_cycles=5)
            res += [np.array(h), mg.get_solution(), st.fine_updates]
            # a single sweep through the operator API flips the finest level's buffers
            mg.set_guess(u1)
            mg.smooth(L, 1)
            st, h = mg.solve(tol=0.0, max_cycles=4)
            res += [np.array(h), mg.get_solution(), st.fine_updates]
            st, h = mg.solve(tol=1e-6, max_cycles=30)
            res += [np.array(h), mg.get_solution(), st.fine_updates, st.cycles]
            # the graph path really ran (1 graph, or 2-3 when the buffer assignment alternates)
            assert mg.graphs_cached() == -1 if graph == "0" else 1 <= mg.graphs_cached() <= 4
        out[graph] = res
    for a, c in zip(out["0"], out["1"]):
        assert np.array_equal(a, c)
    if dt == np.float64 and cfg.get("schedule", 0) == 0:
        _, h_ref = po.Solver(**cfg).solve(b.astype(np.float64), u0.astype(np.float64), tol=0.0, max_cycles=5)
        assert hist_close(out["1"][0], h_ref)


@pytest.mark.parametrize("schedule", [0, 1])
def test_mixed_update_and_residual_in_one_pass_is_bit_identical(pkg, po, monkeypatch, schedule):
    """config 5: k_update_residual (u += s e and r = b - A u, out of place, 32 B per point) against
    the two kernels it replaces (MGX_MIXED_FUSE=0): same double solution, same history"""
    cfg = dict(finest_level=10, coarsest_level=7, mu0=0, mu1=2, mu2=1, schedule=schedule, dtype=2)
    b = po.rhs_sine(10)
    u0 = None if schedule else po.fill_uniform(b.shape, 31)
    out = {}
    for fuse in ("0", "1"):
        monkeypatch.setenv("MGX_MIXED_FUSE", fuse)
        st, h, u = run_gpu(pkg, cfg, b, u0, tol=1e-10, max_cycles=25)
        out[fuse] = (np.array(h), u, st.cycles)
    assert out["0"][2] == out["1"][2]
    assert np.array_equal(out["0"][0], out["1"][0])
    assert np.array_equal(out["0"][1], out["1"][1])
    # and the whole history follows the oracle's (same bottom method: oracle_cfg) to 1e-10
    _, h_ref = po.Solver(**oracle_cfg(po, cfg)).solve(b, u0, tol=1e-10, max_cycles=25)
    assert hist_close(out["1"][0], h_ref)


FULL_SIZE = [
    # BASELINE config 2 exactly: 4096^2, six levels, Jacobi V(2,1)
    ("config2", dict(finest_level=12, coarsest_level=7, mu1=2, mu2=1, schedule=0), 3),
    # the bench's own workload: 8192^2, levels 13..7, the reference's V(10,10)
    ("bench", dict(finest_level=13, coarsest_level=7, mu1=10, mu2=10, schedule=0), 2),
    # BASELINE config 3: 8192^2 red-black Gauss-Seidel
    ("config3", dict(finest_level=13, coarsest_level=7, mu1=2, mu2=1, schedule=0, smoother=1), 2),
]


@pytest.mark.parametrize("name,cfg,cycles", FULL_SIZE, ids=[c[0] for c in FULL_SIZE])
def test_full_size_cycles_against_the_oracle(pkg, po, name, cfg, cycles):
    """BASELINE's full sizes against the oracle itself (its OpenMP build finishes these in
    seconds): residual history to 1e-10 relative per cycle (north_star), iterate to 1e-12 of
    its maximum (the grid operators are bit-exact; the two exact bottom solves differ in the
    last digits)."""
    L = cfg["finest_level"]
    n = (1 << L) - 1
    b = po.rhs_sine(L)
    u0 = po.fill_uniform((n, n), 12345)
    st, h, u = run_gpu(pkg, cfg, b, u0, tol=0.0, max_cycles=cycles)
    u_ref, h_ref = po.Solver(**cfg).solve(b, u0, tol=0.0, max_cycles=cycles)
    assert hist_close(h, h_ref), (h, h_ref)
    assert np.max(np.abs(u - u_ref)) <= 1e-12 * np.max(np.abs(u_ref))


@pytest.mark.parametrize("smoother", [0, 1])
@pytest.mark.parametrize("plan", [("10", "10"), ("8,2", "2,8"), ("2,8", "8,2"), ("10", "5,5"), ("5,5", "10")])
def test_deep_folded_passes_with_the_rhs_window_in_lds_are_bit_identical(pkg, po, monkeypatch, plan, smoother):
    """the 8- and 10-level folded passes keep their rhs delay line in an LDS ring (12-fold unrolled
    step loop, two rows in flight for the restriction variant): every explicit plan that uses them
    must give the bits of the default plan, and the oracle's history"""
    mu = 10 if smoother == 0 else 5            # red-black GS: levels = 2 x sweeps
    pre, post = plan
    if smoother == 1:
        pre, post = [",".join(str(max(1, int(k) // 2)) for k in p.split(",")) for p in (pre, post)]
    cfg = dict(finest_level=11, coarsest_level=8, mu1=mu, mu2=mu, schedule=0, smoother=smoother)
    b = po.rhs_sine(11)
    u0 = po.fill_uniform(b.shape, 4242)
    monkeypatch.setenv("MGX_TILE_MAX_N", "0")
    for k in ("MGX_PLAN_MIN_N", "MGX_PLAN_PRE", "MGX_PLAN_POST"):
        monkeypatch.delenv(k, raising=False)
    _, h0, u_ref = run_gpu(pkg, cfg, b, u0, tol=0.0, max_cycles=2)
    monkeypatch.setenv("MGX_PLAN_MIN_N", "256")
    monkeypatch.setenv("MGX_PLAN_PRE", pre)
    monkeypatch.setenv("MGX_PLAN_POST", post)
    _, h, u = run_gpu(pkg, cfg, b, u0, tol=0.0, max_cycles=2)
    assert np.array_equal(u, u_ref), plan
    assert np.allclose(h, h0, rtol=1e-13, atol=0)
    _, h_orc = po.Solver(**cfg).solve(b, u0, tol=0.0, max_cycles=2)
    assert hist_close(h, h_orc)


def test_arrays_beyond_four_gib_folded_passes_equal_single_sweeps(pkg, monkeypatch):
    """32768^2 in double: 8.6 GB per array, beyond what a 32-bit buffer offset reaches.  The deep folded
    passes address rows through descriptors rebased to each wave's first row (interior stores, every load
    and store of the edge bodies); the single-sweep kernels and stand-alone transfers (MGX_FUSE=1,
    MGX_FOLD=0) use plain 64-bit pointers.  One V(10,10) cycle from device-generated data must give the
    same bits both ways - also the largest grid the library accepts (finest_level 15), on one GPU."""
    cfg = dict(finest_level=15, coarsest_level=7, mu1=10, mu2=10, schedule=0)
    keys = ("MGX_FUSE", "MGX_FOLD", "MGX_TILE_MAX_N")
    out = {}
    for name, env in (("folded", {}), ("single", {"MGX_FUSE": "1", "MGX_FOLD": "0"})):
        for k in keys:
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        with pkg.Multigrid(**cfg) as mg:
            mg.fill_rhs(1, 0.0)
            mg.fill_guess_random(2026)
            st, h = mg.solve(tol=0.0, max_cycles=1)
            # strips of the iterate instead of 8.6 GB twice: the first and last rows (edge bodies), rows
            # around the 4 GiB mark of the array (row 16320 of pitch 32784 doubles) and around the middle
            u = mg.get_solution()
        n = u.shape[0]
        rows = np.r_[0:96, 16256:16448, n // 2 - 64:n // 2 + 64, n - 96:n]
        out[name] = (np.array(h), u[rows].copy(), float(np.abs(u).max()), u[:, ::509].copy())
        del u
    assert np.array_equal(out["folded"][1], out["single"][1])
    assert np.array_equal(out["folded"][3], out["single"][3])
    assert out["folded"][2] == out["single"][2] and out["folded"][2] > 0
    assert np.allclose(out["folded"][0], out["single"][0], rtol=1e-13, atol=0)
    assert out["folded"][0][1] < 0.01 * out["folded"][0][0]          # and the cycle did its work


def test_chunk_geometry_knobs_of_the_deep_passes_never_change_a_bit(pkg, po, monkeypatch):
    """the deep folded passes size their tiles in the launcher (whole rounds of waves, shorter edge tiles,
    edge strips in workgroups of their own): every geometry must give the same bits - uniform tiles,
    other edge ratios, explicit chunk heights incl. ones that leave a last chunk of a few rows"""
    cfg = dict(finest_level=11, coarsest_level=8, mu1=10, mu2=10, schedule=0)
    b = po.rhs_sine(11)
    u0 = po.fill_uniform(b.shape, 99)
    keys = ("MGX_EDGE_SHORT", "MGX_EDGE_PCT", "MGX_LAST_PCT", "MGX_FUSE_ROWS", "MGX_TILE_MAX_N", "MGX_PAIR", "MGX_PAIR_MIN_ROWS",
            "MGX_PAIR_RATIO", "MGX_PAIR_MAX_ROWS", "MGX_MIN_CHUNK")
    ref = None
    # (MGX_PAIR_MIN_ROWS=16: the one-round form with tall chunks on the workgroups dispatched first and short ones on the
    # rest - csrc/mgx_geom.hpp - which the launcher keeps for chunks of 150 rows and more, on these small levels too)
    for env in ({}, {"MGX_EDGE_SHORT": "0"}, {"MGX_EDGE_PCT": "40"}, {"MGX_EDGE_PCT": "8"}, {"MGX_FUSE_ROWS": "36"},
                {"MGX_FUSE_ROWS": "60", "MGX_EDGE_SHORT": "0"}, {"MGX_FUSE_ROWS": "300"}, {"MGX_FUSE_ROWS": "2046"},
                {"MGX_FUSE_ROWS": "1020", "MGX_EDGE_PCT": "45"}, {"MGX_LAST_PCT": "5"}, {"MGX_LAST_PCT": "70", "MGX_EDGE_PCT": "0"},
                {"MGX_FUSE_ROWS": "96", "MGX_LAST_PCT": "50"},
                {"MGX_PAIR_MIN_ROWS": "16"}, {"MGX_PAIR_MIN_ROWS": "16", "MGX_PAIR_RATIO": "200"},
                {"MGX_PAIR_MIN_ROWS": "16", "MGX_PAIR_RATIO": "110", "MGX_EDGE_PCT": "0"},
                {"MGX_PAIR_MIN_ROWS": "16", "MGX_PAIR_RATIO": "170", "MGX_LAST_PCT": "60", "MGX_MIN_CHUNK": "40"},
                {"MGX_PAIR_MIN_ROWS": "16", "MGX_PAIR_MAX_ROWS": "100"}, {"MGX_PAIR": "0"}):
        for k in keys:
            monkeypatch.delenv(k, raising=False)
        monkeypatch.setenv("MGX_TILE_MAX_N", "0")
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        _, h, u = run_gpu(pkg, cfg, b, u0, tol=0.0, max_cycles=2)
        if ref is None:
            ref = (u, h)
            _, h_orc = po.Solver(**cfg).solve(b, u0, tol=0.0, max_cycles=2)
            assert hist_close(h, h_orc)
        assert np.array_equal(u, ref[0]), env
        assert np.allclose(h, ref[1], rtol=1e-13, atol=0), env
