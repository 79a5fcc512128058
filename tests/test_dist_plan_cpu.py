"""The multi-GPU driver's logic on CPU.  The slab plans libmgx emits (mgx_plan_*: the host logic of
csrc/mgx_dist_plan.hpp, the same planner the GPU executor runs) are executed here by world_size-2,
-4 and -8 gloo process groups over numpy slab operators (tests/plan_exec.py), and checked against the
oracle's single-process V-cycle.  Covers the halo exchange, the deep-halo shrinking passes, the slab
restriction / prolongation offsets, the all-gather at the cut level and the all-reduce of the norm.
The numpy operators poison with NaN every row a call does not promise to leave valid, so a plan that
relied on a stale halo row fails."""
import os
import sys

import numpy as np
import pytest

# torch (gloo, CPU tensors) is imported inside the functions that use it, never at module level: `pytest tests -m gpu`
# imports every test module while collecting, and the torch wheel's bundled ROCm runtime must not get into a
# process that is about to run libmgx on the GPU (tests/conftest.py)

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _mgx_cfg(cfg):
    return dict(finest_level=cfg["finest"], coarsest_level=cfg["coarsest"], cut_level=cfg["cut"], mu0=cfg.get("mu0", 0), mu1=cfg["mu1"], mu2=cfg["mu2"],
                omega=cfg["omega"], smoother=1 if cfg["smoother"] == "rbgs" else 0, restrict_mode=cfg["restrict_mode"],
                bottom=cfg["bottom"], schedule=0)


def _worker(rank, world, port, cfg, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist

    import __graft_entry__ as ge
    from dist_cpu_ops import CpuSlabOps, OracleCoarseSolver
    from oracle import pyoracle as po
    from plan_exec import PlanRunner

    pkg = ge.load_package()
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        mcfg = _mgx_cfg(cfg)
        plan = pkg.Plan(world, rank, fold=cfg.get("fold", True), deep=cfg.get("deep", True), **mcfg)
        assert plan.cut == cfg["cut"]
        ops = CpuSlabOps(torch.float64)
        coarse = OracleCoarseSolver(po, cfg["cut"], cfg["coarsest"], cfg)
        mg = PlanRunner(pkg, ops, coarse, plan, mcfg)
        L = cfg["finest"]
        n = (1 << L) - 1
        mg.set_fine("b", np.pad(po.rhs_sine(L), 1))
        if cfg.get("fmg"):
            mg.set_fine("u", np.zeros((n + 2, n + 2)))
        else:
            mg.set_fine("u", np.pad(po.fill_uniform((n, n), 12345), 1))
        k, hist = mg.solve(tol=1e-8, max_cycles=cfg["max_cycles"], fmg=bool(cfg.get("fmg")))
        if rank == 0:
            ret["hist"] = hist
        ret[f"rows{rank}"] = mg.own_interior()
        ret[f"exch{rank}"] = mg.exchanges
    finally:
        dist.destroy_process_group()


def _run(world, cfg):
    import torch.multiprocessing as mp

    mgr = mp.Manager()
    ret = mgr.dict()
    port = 29500 + (os.getpid() % 2000) + world
    mp.spawn(_worker, args=(world, port, cfg, ret), nprocs=world, join=True)
    return dict(ret)


def _reference(po, cfg):
    L = cfg["finest"]
    n = (1 << L) - 1
    s = po.Solver(finest_level=L, coarsest_level=cfg["coarsest"], mu0=cfg.get("mu0", 0), mu1=cfg["mu1"], mu2=cfg["mu2"], omega=cfg["omega"],
                  smoother=1 if cfg["smoother"] == "rbgs" else 0, schedule=1 if cfg.get("fmg") else 0,
                  restrict_mode=cfg["restrict_mode"], bottom=cfg["bottom"])
    u0 = None if cfg.get("fmg") else po.fill_uniform((n, n), 12345)
    return s.solve(po.rhs_sine(L), u0, tol=1e-8, max_cycles=cfg["max_cycles"])


BASE = dict(finest=8, cut=6, coarsest=4, mu1=2, mu2=1, omega=2.0 / 3.0, smoother="jacobi", restrict_mode=0, bottom=0,
            max_cycles=6)


@pytest.mark.parametrize("fold", [True, False])
@pytest.mark.parametrize("world", [2, 4])
@pytest.mark.parametrize("smoother,mu1,mu2", [("jacobi", 2, 1), ("jacobi", 3, 3), ("rbgs", 1, 1), ("jacobi", 2, 0),
                                              ("jacobi", 0, 2)])
def test_slab_plans_match_the_single_process_oracle(po, world, smoother, mu1, mu2, fold):
    """fold = True: transfers folded into the smoother passes (CYCLE operations, the default);
    False: separate SMOOTH / RESTRICT / PROLONG / SUMSQ operations"""
    cfg = dict(BASE, smoother=smoother, mu1=mu1, mu2=mu2, fold=fold)
    got = _run(world, cfg)
    u_ref, h_ref = _reference(po, cfg)
    h = np.array(got["hist"])
    assert len(h) == len(h_ref)
    assert np.all(np.abs(h - h_ref) <= 1e-10 * h_ref + 1e-13 * h_ref[0]), (h, h_ref)
    for r in range(world):
        lo, hi, own = got[f"rows{r}"]
        assert np.max(np.abs(own - u_ref[lo - 1:hi - 1])) <= 1e-12 * np.max(np.abs(u_ref))
    # communication-avoiding plan: one u exchange per cycle on the finest level plus ONE exchange
    # (the restricted right-hand side) per further distributed level; the correction needs none
    cycles = len(h) - 1
    levels = cfg["finest"] - cfg["cut"]
    assert got["exch0"] <= cycles * (1 + (levels - 1)) + 2, got["exch0"]


@pytest.mark.parametrize("deep", [True, False])
def test_three_distributed_levels_and_replicated_coarse(po, deep):
    """deep: every level below the finest leaves its correction valid as far into the halos as
    the level above reads it (3 exchanges per cycle for three slab levels); not deep: the
    correction is exchanged (5 per cycle)"""
    cfg = dict(BASE, finest=9, cut=6, coarsest=5, mu1=2, mu2=2, max_cycles=4, deep=deep)
    got = _run(2, cfg)
    u_ref, h_ref = _reference(po, cfg)
    h = np.array(got["hist"])
    assert np.all(np.abs(h - h_ref) <= 1e-10 * h_ref + 1e-13 * h_ref[0])
    per_cycle = (1 + 2) if deep else (1 + 2 * 2)
    assert got["exch0"] <= (len(h) - 1) * per_cycle + 2, got["exch0"]
    if not deep:
        assert got["exch0"] > (len(h) - 1) * (1 + 2) + 2


def test_eight_ranks_three_distributed_levels(po):
    """the round-end layout in miniature: 8 ranks, three slab levels above a replicated cut
    level (bench.py --gpus 8 at 16384^2 distributes levels 14..11 and replicates <= 10)"""
    cfg = dict(BASE, finest=10, cut=7, coarsest=5, mu1=2, mu2=2, max_cycles=3)
    got = _run(8, cfg)
    u_ref, h_ref = _reference(po, cfg)
    h = np.array(got["hist"])
    assert len(h) == len(h_ref)
    assert np.all(np.abs(h - h_ref) <= 1e-10 * h_ref + 1e-13 * h_ref[0]), (h, h_ref)
    rows = 0
    for r in range(8):
        lo, hi, own = got[f"rows{r}"]
        rows += hi - lo
        assert np.max(np.abs(own - u_ref[lo - 1:hi - 1])) <= 1e-12 * np.max(np.abs(u_ref))
    assert rows == (1 << 10) - 1                       # the slabs tile the unknown rows exactly
    assert got["exch0"] <= (len(h) - 1) * (1 + 2) + 2, got["exch0"]


@pytest.mark.parametrize("world,mu0,smoother,fold", [(2, 0, "jacobi", True), (4, 1, "jacobi", True), (2, 1, "rbgs", True),
                                                    (2, 0, "jacobi", False)])
def test_fullmultigrid_on_slabs_matches_the_oracle(po, world, mu0, smoother, fold):
    """PS:629-650 through the slab plan (mgx_plan_fmg): right-hand sides restricted level by level with
    their halos exchanged, fullmultigrid on the replicated levels, prolongation + mu0 + 1 V-cycles per
    slab level; then V-cycles to the tolerance - the oracle's FMG solve"""
    cfg = dict(BASE, finest=9, cut=6, coarsest=4, mu0=mu0, mu1=2, mu2=1, smoother=smoother, fold=fold, fmg=True, max_cycles=5)
    got = _run(world, cfg)
    u_ref, h_ref = _reference(po, cfg)
    h = np.array(got["hist"])
    assert len(h) == len(h_ref)
    assert np.all(np.abs(h - h_ref) <= 1e-10 * h_ref + 1e-13 * h_ref[0]), (h, h_ref)
    for r in range(world):
        lo, hi, own = got[f"rows{r}"]
        assert np.max(np.abs(own - u_ref[lo - 1:hi - 1])) <= 1e-12 * np.max(np.abs(u_ref))


def test_plan_geometry_and_defaults(pkg):
    """the planner alone (no process group): the bench's own 8-GPU layout at 16384^2"""
    cfgs = dict(finest_level=14, coarsest_level=7, mu1=10, mu2=10)
    plans = [pkg.Plan(8, g, **cfgs) for g in range(8)]
    assert all(p.cut == 10 for p in plans)             # 2048^2 and up distributed, <= 1024^2 replicated
    for l in range(11, 15):
        N = 1 << l
        geo = [p.level(l) for p in plans]
        assert geo[0].own_lo == 0 and geo[-1].own_hi == N + 1
        for a, b in zip(geo, geo[1:]):
            assert a.own_hi == b.own_lo                # the slabs tile rows 0..N
        for g in geo:
            assert g.row0 == max(g.own_lo - g.halo, 0) and g.rows == min(g.own_hi + g.halo, N + 1) - g.row0
            assert N // 8 >= g.halo
    # every slab emits the same sequence of operation codes (the one-process executor relies on it)
    for p in plans:
        p.guess_set(True)
    for _ in range(2):
        seqs = [[(o.op, o.level, o.depth) for o in p.vcycle()] for p in plans]
        assert all(s == seqs[0] for s in seqs)
        ex = [o for o in seqs[0] if o[0] == pkg.binding.DOP_EXCHANGE]
        assert len(ex) <= 4                            # one per distributed level
        assert all([o.op for o in p.norm()] == [pkg.binding.DOP_ALLREDUCE_NORM] for p in plans)
    # slabs too thin for their halos are refused with a message, not mis-planned
    with pytest.raises(pkg.MgxError, match="raise cut_level"):
        pkg.Plan(8, 0, finest_level=9, coarsest_level=5, cut_level=6, mu1=10, mu2=10)
