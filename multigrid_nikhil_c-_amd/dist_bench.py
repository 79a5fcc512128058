"""bench.py's N > 1 leg: one process per GPU (torch.distributed.run), RCCL halo
exchange, the same 8192^2 problem split into row slabs (strong scaling)."""
from __future__ import annotations

import json
import os
import time

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import torch
import torch.distributed as dist

from . import binding as B
from .dist import DistMultigrid, HipCoarseSolver, HipSlabOps

HBM_PEAK_GBS = 8000.0


def _hash_uniform(rows, cols, N, seed=12345):
    """u0 ~ U(-1,1) from the global node index (independent of the decomposition)"""
    idx = (rows.to(torch.int64) - 1) * (N - 1) + (cols.to(torch.int64) - 1) + seed * 1000003
    x = idx * 6364136223846793005 + 1442695040888963407
    x = x ^ (x >> 29)
    x = x * (-4658895280553007687)
    x = x ^ (x >> 32)
    return ((x >> 11) & ((1 << 52) - 1)).to(torch.float64) / float(1 << 51) - 1.0


def _rhs_sine(rows, cols, N):
    h = 1.0 / N
    x = cols.to(torch.float64) * h
    y = rows.to(torch.float64) * h
    return h * h * 8.0 * torch.pi ** 2 * torch.sin(2 * torch.pi * x) * torch.sin(2 * torch.pi * y)


def default_cut(finest, coarsest, world, halo):
    """levels up to 2048^2 are replicated: below that a replicated level costs
    less than the exchanges a distributed one needs (DESIGN.md "Multi-GPU")"""
    cut = max(coarsest, min(finest - 2, 11))
    while cut < finest - 1 and ((1 << (cut + 1)) // world) < 2 * halo:
        cut += 1
    return cut


def single_gpu_reference(args, L, steps):
    """rank 0 only: the same workload on ONE GPU through the ordinary handle
    (untimed part of the job; gives the like-for-like strong-scaling baseline)"""
    import time as _t

    mg = B.Multigrid(finest_level=L, coarsest_level=min(args.coarsest, L), mu0=0, mu1=args.mu1, mu2=args.mu2,
                     omega=args.omega, smoother=B.SMOOTHER_RBGS if args.smoother == "rbgs" else B.SMOOTHER_JACOBI,
                     dtype=B.DTYPE_F32 if args.dtype == "f32" else B.DTYPE_F64, schedule=B.SCHEDULE_V,
                     device=torch.cuda.current_device())
    try:
        mg.fill_rhs(1, 0.0)
        mg.fill_guess_random(12345)
        mg.solve(tol=0.0, max_cycles=1)
        mg.synchronize()
        t0 = _t.perf_counter()
        st, _ = mg.solve(tol=0.0, max_cycles=steps)
        mg.synchronize()
        secs = _t.perf_counter() - t0
        return {"value": st.fine_updates / secs, "unit": "updates/s", "ms_per_step": secs / steps * 1e3, "steps": steps}
    finally:
        mg.close()


def run(args, emit=None):
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    # Rehearsal mode for a 1-GPU box (RCCL refuses two ranks on one device): backend gloo with
    # host-staged halos, every rank on device 0.  The default is what the contract asks for:
    # one rank per GPU, backend nccl (= RCCL over xGMI).
    backend = os.environ.get("MGX_DIST_BACKEND", "nccl")
    if os.environ.get("MGX_DIST_SINGLE_DEVICE"):
        local = 0
    torch.cuda.set_device(local)
    use_pg = world > 1 or bool(os.environ.get("MGX_FORCE_DIST"))
    if use_pg:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend)
    dtype = torch.float32 if args.dtype == "f32" else torch.float64
    if args.dtype == "mixed":
        raise SystemExit("--dtype mixed is a single-GPU configuration in this round")
    per = 2 if args.smoother == "rbgs" else 1
    halo = per * args.mu1 + max(per * args.mu2, 2)
    L = args.level
    ref = None
    if rank == 0 and world > 1:
        ref = single_gpu_reference(args, L, max(1, min(args.steps, 5)))
    torch.cuda.synchronize()
    cut = default_cut(L, args.coarsest, world, halo)
    cfg = dict(mu1=args.mu1, mu2=args.mu2, omega=args.omega, smoother=args.smoother,
               restrict_mode=B.RESTRICT_CONSISTENT, bottom=B.BOTTOM_EXACT)
    ops = HipSlabOps(dtype)
    coarse = HipCoarseSolver(cut, min(args.coarsest, cut), cfg, dtype)
    mg = DistMultigrid(ops, coarse, L, cut, mu1=args.mu1, mu2=args.mu2, omega=args.omega, smoother=args.smoother,
                       restrict_mode=cfg["restrict_mode"], staged_halo=(backend != "nccl"))
    mg.profile = True
    mg.set_fine("b", _rhs_sine)
    mg.set_fine("u", _hash_uniform)
    n = (1 << L) - 1

    def barrier():
        torch.cuda.synchronize()
        if use_pg:
            dist.barrier()
        torch.cuda.synchronize()

    # the "V-cycles to 1e-8" half of the metric (untimed)
    k_tol, hist0 = mg.solve(tol=1e-8, max_cycles=60)
    mg.set_fine("u", _hash_uniform)
    for _ in range(args.warmup):
        mg.vcycle()
        mg.residual_norm()
    mg.fine_updates = 0.0
    mg.reset_profile()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        mg.vcycle()
        mg.residual_norm()
    barrier()
    secs = time.perf_counter() - t0
    t = torch.tensor([secs], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
    if use_pg:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    secs = float(t.item())
    prof = mg.collect_profile()
    if rank == 0:
        es = 4 if args.dtype == "f32" else 8
        achieved = prof["bytes"] / (prof["ms"] * 1e-3) / 1e9 if prof["ms"] > 0 else 0.0
        out = {
            "metric": "fine_grid_stencil_updates_per_sec",
            "value": mg.fine_updates / secs,
            "unit": "updates/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": secs / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": args.dtype,
            "data": "synthetic",
            "config": {
                "workload": f"2D Poisson {1 << L}^2 (n={n} interior), {L - min(args.coarsest, cut) + 1}-level V({args.mu1},{args.mu2}) cycle, "
                            f"{'weighted Jacobi w=%.4f' % args.omega if args.smoother == 'jacobi' else 'red-black Gauss-Seidel'}, "
                            f"{args.dtype}, row slabs over {world} GPUs on levels {cut + 1}..{L} ({mg.halo}-row deep halos on the finest level, "
                            f"{'RCCL send/recv' if backend == 'nccl' else backend + ' with host-staged halos (rehearsal)'}, "
                            f"{'transfers folded into the smoother passes' if mg.fold else 'separate transfer kernels'}), "
                            f"levels <= {cut} replicated, exact bottom solve at {(1 << min(args.coarsest, cut)) - 1}^2",
                "finest_level": L, "coarsest_level": min(args.coarsest, cut), "cut_level": cut, "mu1": args.mu1,
                "mu2": args.mu2, "smoother": args.smoother, "step": "one V-cycle + residual norm",
                "parallelism": f"slab{world}",
            },
            "vcycles_to_1e-8": k_tol if hist0[-1] <= 1e-8 * hist0[0] else None,
            "single_gpu_same_workload": ref,
            "speedup_vs_single_gpu_same_workload": (mg.fine_updates / secs / ref["value"]) if ref else None,
            "halo_exchanges_per_step": mg.exchanges_timed / max(args.steps, 1),
            "roofline": {
                "bound": "hbm",
                "kernel": ("k_jacobi_fused / k_jacobi_cycle<%s,K,PRE,POST,%d> on the slab's row window (deep halos, "
                           "transfers folded into the passes)" if mg.fold else
                           "k_jacobi_fused<%s,K,%d> (slab rows, deep halos; separate transfer kernels)")
                          % ("double" if es == 8 else "float", 1 if args.smoother == "rbgs" else 0),
                "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                "traffic": None,
                "how": "rank 0: torch.cuda events (current stream = the kernels' stream) around every finest-level "
                       "smoothing block in the timed steps; bytes = 3*sizeof(T) per point actually updated, halo rows "
                       "recomputed by the deep-halo scheme included",
                "launches_timed": prof["launches"],
            },
        }
        if emit is not None:
            emit(out)
        else:
            print(json.dumps(out), flush=True)
    if use_pg:
        dist.barrier()
        dist.destroy_process_group()
