"""bench.py's N > 1 leg: one process per GPU, each rank a libmgx multi-GPU handle
(mgx_create_rank: csrc/mgx_dist.hpp) owning one row slab of BASELINE config 4's grid; halo rows move by
RCCL send/recv over xGMI on the slab's stream (built-in transport).  The control plane is the TCP store
of rendezvous.py: it hands rank 0's ncclUniqueId to the other ranks and carries the barrier and the
max-over-ranks of the timed region the bench contract asks for.  No torch in a rank process: libmgx runs
on the ROCm stack it was built against (the line reports the mapped libraries, `runtime_libs`).

If the RCCL transport fails on any rank the job reports `value: null` with the error and exits non-zero:
a host-staged number is never printed under an N-GPU label.  Rehearsal on a 1-GPU box (RCCL refuses two
ranks on one device) is an explicit opt-in: MGX_DIST_SINGLE_DEVICE=1 MGX_DIST_BACKEND=staged puts every
rank on device 0 and swaps the RCCL transport for the host-staged one (transport.py) - same C++
executor, plans and kernels; the line then carries `transport: "staged"` and its timing means nothing."""
from __future__ import annotations

import json
import os
import sys
import time

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

from . import binding as B
from .rendezvous import Store
from .transport import StagedTransport, broadcast_rccl_id

HBM_PEAK_GBS = 8000.0


def single_gpu_reference(args, L, steps, device):
    """rank 0 only: the same workload on ONE GPU through the ordinary handle
    (untimed part of the job; gives the like-for-like strong-scaling baseline)"""
    mg = B.Multigrid(finest_level=L, coarsest_level=min(args.coarsest, L), mu0=0, mu1=args.mu1, mu2=args.mu2,
                     omega=args.omega, smoother=B.SMOOTHER_RBGS if args.smoother == "rbgs" else B.SMOOTHER_JACOBI,
                     dtype=B.DTYPE_F32 if args.dtype == "f32" else B.DTYPE_F64, schedule=B.SCHEDULE_V, device=device,
                     arith=B.ARITH_FMA if getattr(args, "arith", "fma") == "fma" else B.ARITH_SEPARATE)
    try:
        mg.fill_rhs(1, 0.0)
        mg.fill_guess_random(12345)
        mg.solve(tol=0.0, max_cycles=1)
        mg.synchronize()
        t0 = time.perf_counter()
        st, _ = mg.solve(tol=0.0, max_cycles=steps)
        mg.synchronize()
        secs = time.perf_counter() - t0
        return {"value": st.fine_updates / secs, "unit": "updates/s", "ms_per_step": secs / steps * 1e3, "steps": steps}
    finally:
        mg.close()


def null_line(args, world, error, transport):
    L = args.level
    return {"metric": "fine_grid_stencil_updates_per_sec", "value": None, "unit": "updates/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": args.dtype, "data": "synthetic", "transport": transport,
            "config": {"workload": f"2D Poisson {1 << L}^2, row slabs over {world} GPUs"}, "error": error}


def run(args, emit=None):
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    backend = os.environ.get("MGX_DIST_BACKEND", "rccl")          # data plane: "rccl" (= "nccl") | "staged" (= "gloo"), explicit
    backend = {"nccl": "rccl", "gloo": "staged"}.get(backend, backend)
    if backend not in ("rccl", "staged"):
        raise SystemExit(f"MGX_DIST_BACKEND={backend!r}: expected rccl or staged")
    rehearsal = bool(os.environ.get("MGX_DIST_SINGLE_DEVICE"))
    if rehearsal:
        local = 0
    if args.dtype == "mixed":
        raise SystemExit("--dtype mixed is a single-GPU configuration")
    out_line = emit if emit is not None else (lambda o: print(json.dumps(o), flush=True))
    # control plane (every blocking call has a deadline: rendezvous.py)
    store = Store(rank, world)
    assert world == args.gpus, (world, args.gpus)
    L = args.level
    n = (1 << L) - 1
    cfg = dict(finest_level=L, coarsest_level=min(args.coarsest, L), mu0=0, mu1=args.mu1, mu2=args.mu2, omega=args.omega,
               smoother=B.SMOOTHER_RBGS if args.smoother == "rbgs" else B.SMOOTHER_JACOBI,
               dtype=B.DTYPE_F32 if args.dtype == "f32" else B.DTYPE_F64, schedule=B.SCHEDULE_V, device=local, profile=1,
               arith=B.ARITH_FMA if getattr(args, "arith", "fma") == "fma" else B.ARITH_SEPARATE)
    ref = None
    if rank == 0:
        ref = single_gpu_reference(args, L, max(1, min(args.steps, 5)), local)
    store.barrier()
    transport = None
    mg = None
    if backend == "rccl":
        # the built-in RCCL transport; one warm-up cycle proves every collective it uses before anything is
        # timed.  A failure on ANY rank ends the job with value: null (never a host-staged number under this label).
        err = ""
        try:
            rccl_id = broadcast_rccl_id(store)
            mg = B.Multigrid.rank(rank, world, rccl_id=rccl_id, **cfg)
            mg.fill_rhs(1, 0.0)
            mg.fill_guess_random(12345)
            mg.solve(tol=0.0, max_cycles=1)
            mg.synchronize()
        except Exception as e:      # noqa: BLE001 - whatever it is, the other ranks must hear of it
            err = f"rank {rank}: {type(e).__name__}: {e}"
        errs = [e.decode() for e in store.allgather(err.encode()) if e]
        if errs:
            if rank == 0:
                out_line(null_line(args, world, "RCCL transport failed: " + " | ".join(errs), "rccl"))
            sys.stderr.write(f"dist_bench: RCCL transport failed ({'; '.join(errs)})\n")
            # no mgx_destroy on a communicator whose peers may be gone (it is aborted at process exit)
            os._exit(4)
        wire = "RCCL send/recv (built-in transport, ncclCommInitRank over xGMI)"
    else:
        transport = StagedTransport(store)
        mg = B.Multigrid.rank(rank, world, transport=transport.struct, **cfg)
        wire = "host-staged halos through the TCP store (rehearsal transport, explicit MGX_DIST_BACKEND=staged)"
    plan = B.Plan(world, rank, **{k: v for k, v in cfg.items() if k not in ("device", "profile")})
    cut = plan.cut
    halo = plan.level(L).halo

    def barrier():
        mg.synchronize()
        store.barrier()

    mg.fill_rhs(1, 0.0)
    mg.fill_guess_random(12345)
    # the "V-cycles to 1e-8" half of the metric (untimed)
    st0, hist0 = mg.solve(tol=1e-8, max_cycles=60)
    mg.fill_guess_random(12345)
    if args.warmup > 0:
        mg.solve(tol=0.0, max_cycles=args.warmup)
    mg.profile_reset()
    ex0 = mg.exchanges()
    barrier()
    t0 = time.perf_counter()
    st, hist = mg.solve(tol=0.0, max_cycles=args.steps)       # exactly K x (one V-cycle + residual norm)
    barrier()
    secs = store.allreduce_max(time.perf_counter() - t0)
    ranks_seen = store.allreduce_sum_int(1)
    prof = mg.profile()
    exchanges = mg.exchanges() - ex0
    assert st.cycles == args.steps
    if rank == 0:
        if ranks_seen != args.gpus:
            raise SystemExit(f"--gpus {args.gpus} but the collective saw {ranks_seen} ranks")
        es = 4 if args.dtype == "f32" else 8
        # rank 0's finest-level smoother passes: each launch must move its slab rows once
        # (read v, read b, write v'), halo rows recomputed by the deep-halo scheme included
        g = plan.level(L)
        rows = g.rows
        sm_ms, sm_launches, sm_sweeps = prof["ms"][0], max(prof["launches"][0], 1), prof["sweeps"][0]
        avg_ms = sm_ms / sm_launches
        bytes_per_launch = 3.0 * es * rows * n
        achieved = bytes_per_launch / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
        out = {
            "metric": "fine_grid_stencil_updates_per_sec",
            "value": st.fine_updates / secs,
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
                "workload": f"2D Poisson {1 << L}^2 (n={n} interior), {L - cfg['coarsest_level'] + 1}-level V({args.mu1},{args.mu2}) cycle, "
                            f"{'weighted Jacobi w=%.4f' % args.omega if args.smoother == 'jacobi' else 'red-black Gauss-Seidel'}, "
                            f"{args.dtype}, row slabs over {world} GPUs on levels {cut + 1}..{L} ({halo}-row deep halos on the finest "
                            f"level, {wire}, transfers folded into the smoother passes), levels <= {cut} replicated, exact "
                            f"bottom solve at {(1 << cfg['coarsest_level']) - 1}^2",
                "finest_level": L, "coarsest_level": cfg["coarsest_level"], "cut_level": cut, "mu1": args.mu1,
                "mu2": args.mu2, "smoother": args.smoother, "arith": getattr(args, "arith", "fma"),
                "step": "one V-cycle + residual norm (mgx_solve loop body)",
                "parallelism": f"slab{world}", "driver": "C++ (mgx_create_rank), one process per GPU",
                "rehearsal_all_ranks_on_one_device": rehearsal,
            },
            "ranks_seen_by_collective": ranks_seen,
            "transport": backend,
            "runtime_libs": B.runtime_libs(),
            "vcycles_to_1e-8": st0.cycles if st0.converged else None,
            "single_gpu_same_workload": ref,
            "speedup_vs_single_gpu_same_workload": (st.fine_updates / secs / ref["value"]) if ref else None,
            "halo_exchanges_per_step": exchanges / max(args.steps, 1),
            "roofline": {
                "bound": "hbm",
                "kernel": "k_jacobi_fused / k_jacobi_cycle<%s,K,PRE,POST,%d> on rank 0's slab of the finest level (deep halos, "
                          "transfers folded into the passes)" % ("double" if es == 8 else "float", 1 if args.smoother == "rbgs" else 0),
                "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                "traffic": None,
                "algorithmic_bytes_per_launch": bytes_per_launch, "avg_launch_ms": avg_ms, "launches_timed": sm_launches,
                "sweeps_per_pass": sm_sweeps / sm_launches,
                "how": "rank 0: HIP events on the slab's stream around every finest-level smoothing block in the timed steps; "
                       "bytes = 3*sizeof(T) x (slab rows incl. halos) x n per pass, charged once per pass",
            },
        }
        out_line(out)
    store.barrier()
    mg.close()
    store.close()
