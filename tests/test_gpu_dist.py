"""The multi-GPU driver over the real HIP slab operators on one MI355X:
world_size 1 (no communication) must reproduce mgx_solve bit for bit, and two
ranks sharing the GPU (gloo with host-staged halos, because RCCL refuses two
ranks on one device) must reproduce it too."""
import os
import sys

import numpy as np
import pytest
import torch  # noqa: F401  (before libmgx.so is loaded: see HipSlabOps.__init__ on the two HIP runtimes)

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _setup(pkg, po, cfg, staged=False, fold=None):
    import torch

    from multigrid_nikhil_c_amd.dist import DistMultigrid, HipCoarseSolver, HipSlabOps

    dt = torch.float32 if cfg.get("dtype") == "f32" else torch.float64
    ccfg = dict(mu1=cfg["mu1"], mu2=cfg["mu2"], omega=cfg["omega"], smoother=cfg["smoother"], restrict_mode=0, bottom=0)
    mg = DistMultigrid(HipSlabOps(dt), HipCoarseSolver(cfg["cut"], cfg["coarsest"], ccfg, dt), cfg["finest"], cfg["cut"],
                       mu1=cfg["mu1"], mu2=cfg["mu2"], omega=cfg["omega"], smoother=cfg["smoother"], staged_halo=staged,
                       fold=cfg.get("fold", fold), deep=cfg.get("deep"))
    L = cfg["finest"]
    n = (1 << L) - 1
    npdt = np.float32 if dt == torch.float32 else np.float64
    b = po.rhs_sine(L).astype(npdt)
    u0 = po.fill_uniform((n, n), 12345).astype(npdt)
    bt, ut = torch.from_numpy(np.pad(b, 1)).cuda(), torch.from_numpy(np.pad(u0, 1)).cuda()
    mg.set_fine("b", lambda r, c, N: bt[r, c])
    mg.set_fine("u", lambda r, c, N: ut[r, c])
    return mg, b, u0


def _single(pkg, cfg, b, u0, cycles):
    with pkg.Multigrid(finest_level=cfg["finest"], coarsest_level=cfg["coarsest"], mu1=cfg["mu1"], mu2=cfg["mu2"],
                       omega=cfg["omega"], smoother=1 if cfg["smoother"] == "rbgs" else 0, schedule=0,
                       dtype=0 if cfg.get("dtype") == "f32" else 1) as mg:
        mg.set_rhs(b)
        mg.set_guess(u0)
        st, h = mg.solve(tol=0.0, max_cycles=cycles)
        return h, mg.get_solution()


@pytest.mark.parametrize("fold", [True, False])
@pytest.mark.parametrize("smoother,mu1,mu2", [("jacobi", 2, 1), ("jacobi", 10, 10), ("rbgs", 2, 1), ("jacobi", 3, 0),
                                              ("jacobi", 0, 2), ("rbgs", 1, 3)])
def test_world1_slab_driver_equals_single_gpu_solve(pkg, po, smoother, mu1, mu2, fold):
    """fold: the transfers ride on the smoother passes (mgx_slab_cycle, the default) or run as
    separate slab kernels; either way the same bits as mgx_solve"""
    cfg = dict(finest=10, cut=8, coarsest=6, mu1=mu1, mu2=mu2, omega=2.0 / 3.0, smoother=smoother, fold=fold)
    mg, b, u0 = _setup(pkg, po, cfg)
    hist = [mg.residual_norm()]
    for _ in range(3):
        mg.vcycle()
        hist.append(mg.residual_norm())
    h_ref, u_ref = _single(pkg, cfg, b, u0, 3)
    assert np.array_equal(mg.own_interior("u").cpu().numpy(), u_ref)       # same kernels, same order: bitwise
    assert np.allclose(hist, h_ref, rtol=1e-13, atol=0)
    _, h_orc = po.Solver(finest_level=10, coarsest_level=6, mu1=mu1, mu2=mu2, smoother=1 if smoother == "rbgs" else 0,
                         schedule=0).solve(b, u0, tol=0.0, max_cycles=3)
    assert np.all(np.abs(np.array(hist) - h_orc) <= 1e-10 * h_orc + 1e-13 * h_orc[0])


def _worker(rank, world, port, cfg, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      HSA_ENABLE_IPC_MODE_LEGACY="0")
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist

    import __graft_entry__ as ge
    from oracle import pyoracle as po

    pkg = ge.load_package()
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        mg, b, u0 = _setup(pkg, po, cfg, staged=True)
        hist = [mg.residual_norm()]
        for _ in range(cfg["cycles"]):
            mg.vcycle()
            hist.append(mg.residual_norm())
        lv = mg.lv[cfg["finest"]]
        ret[f"rows{rank}"] = (max(lv.own_lo, 1), min(lv.own_hi, lv.N), mg.own_interior("u").cpu().numpy())
        if rank == 0:
            ret["hist"] = hist
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,smoother,mu1,mu2,fold", [
    (2, "jacobi", 3, 2, True), (2, "rbgs", 1, 1, True), (2, "jacobi", 10, 10, True), (2, "jacobi", 3, 2, False),
    # several passes per block with the transfers folded into the first / last one
    (2, "jacobi", 7, 6, True), (2, "rbgs", 3, 3, True), (2, "jacobi", 10, 10, False), (2, "jacobi", 8, 9, True),
    (2, "jacobi", 6, 7, True), (2, "rbgs", 4, 5, True),
    # interior ranks have two slab edges
    (4, "jacobi", 10, 10, True), (4, "rbgs", 2, 1, True),
    # float slabs (world < 0 marks them)
    (-2, "jacobi", 10, 10, True), (-2, "rbgs", 2, 2, True), (-2, "jacobi", 4, 3, False)])
def test_ranks_sharing_one_gpu_equal_single_gpu_solve(pkg, po, world, smoother, mu1, mu2, fold):
    import torch.multiprocessing as mp

    dtype = "f32" if world < 0 else "f64"
    world = abs(world)
    big = mu1 >= 7 or world > 2
    cfg = dict(finest=10 if big else 9, cut=7, coarsest=5, mu1=mu1, mu2=mu2, omega=2.0 / 3.0, smoother=smoother,
               cycles=3, fold=fold, dtype=dtype, deep=not (mu1 == 3 and mu2 == 2))   # (3,2): the correction is exchanged
    ret = mp.Manager().dict()
    mp.spawn(_worker, args=(world, 29700 + os.getpid() % 1000, cfg, ret), nprocs=world, join=True)
    n = (1 << cfg["finest"]) - 1
    npdt = np.float32 if dtype == "f32" else np.float64
    b, u0 = po.rhs_sine(cfg["finest"]).astype(npdt), po.fill_uniform((n, n), 12345).astype(npdt)
    h_ref, u_ref = _single(pkg, cfg, b, u0, 3)
    assert np.allclose(ret["hist"], h_ref, rtol=1e-13, atol=0)
    for r in range(world):
        lo, hi, own = ret[f"rows{r}"]
        assert np.array_equal(own, u_ref[lo - 1:hi - 1])


def test_poisson_driver_binary_runs_the_reference_sequence(pkg):
    """the C++ driver shaped like PS:658-731 (build levels, load vector,
    fullmultigrid, print size) on the reference's own hierarchy 10..7"""
    import subprocess

    exe = os.path.join(ROOT, "multigrid_nikhil_c-_amd", "host", "poisson_driver")
    if not os.path.exists(exe):
        subprocess.run(["make", "-C", os.path.dirname(exe)], check=True)
    out = subprocess.run([exe, "10", "7", "0", "2", "1"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr
    assert "Size of finest level solution is 1046529" in out.stdout        # PS:728, 1023^2
    assert "Program Running Correctly" in out.stdout                       # PS:729
    line = [ln for ln in out.stdout.splitlines() if ln.startswith("solve to 1e-8")][0]
    assert "u(1/2,1/2) = 0.29468" in out.stdout
    assert int(line.split()[3]) <= 16


def test_config4_grid_slab_driver_equals_single_gpu_solve(pkg, po):
    """BASELINE config 4's grid (16384^2, levels 14..7) through the slab driver exactly as
    bench.py --gpus N sets it up (levels 14..12 on slabs, <= 11 in the replicated handle),
    world 1 on this one GPU: the reference's V(10,10) cycle must reproduce mgx_solve bit for bit"""
    import gc

    cfg = dict(finest=14, cut=11, coarsest=7, mu1=10, mu2=10, omega=2.0 / 3.0, smoother="jacobi")
    mg, b, u0 = _setup(pkg, po, cfg)
    hist = [mg.residual_norm()]
    mg.vcycle()
    hist.append(mg.residual_norm())
    own = mg.own_interior("u").cpu().numpy()
    del mg
    gc.collect()
    h_ref, u_ref = _single(pkg, cfg, b, u0, 1)
    assert np.array_equal(own, u_ref)
    assert np.allclose(hist, h_ref, rtol=1e-13, atol=0)
    assert hist[1] < 0.1 * hist[0]
