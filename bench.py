#!/usr/bin/env python3
"""bench.py — headline benchmark of the 2-D Poisson multigrid hot path on MI355X.

Metric (BASELINE.json): fine-grid stencil updates/sec (+ V-cycles to 1e-8
residual) on the 8192^2 Poisson problem.

A "step" is one V-cycle of the whole hierarchy over the resident problem
(smoothing on every level, fused residual+restriction, prolongation+correction,
exact bottom solve) followed by the residual-norm evaluation the solve loop
makes after every cycle: mgx_solve(tol=0, max_cycles=K) runs exactly K of them.
`value` = finest-level smoother point updates / wall time of those K steps,
i.e. the coarse levels, the transfers, the bottom solve and the norm are all
inside the time but only fine-grid updates are counted.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--level L] [--dtype f64|f32|mixed]
                    [--smoother jacobi|rbgs] [--mu1 A --mu2 B] [--no-cpu-baseline]

N = 1 : one process, libmgx (C-ABI) through ctypes, HIP events for the roofline.
N > 1 : one process per GPU (ranks from torch.distributed.run or from this script's own launcher;
        RCCL inside libmgx, a small TCP store as control plane - no torch in a rank process), the
        finest levels slab-decomposed by rows with halo exchange between smoothing blocks
        (csrc/mgx_dist.hpp, multigrid_nikhil_c-_amd/dist_bench.py).  Default grid 16384^2 (BASELINE config 4,
        the north star's strong-scaling grid), the same at every N > 1; rank 0 also
        times the single-GPU solver on that same grid inside the job and reports it
        (`single_gpu_same_workload`) so the strong-scaling factor is like for like.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC (RCCL / multi-process GPU work)

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# Native libraries (RCCL prints a version banner) write to fd 1; the contract is
# ONE JSON line on stdout, so everything else is routed to stderr and the result
# line goes to the saved descriptor.
_REAL_STDOUT = os.dup(1)
os.dup2(2, 1)


def emit(obj):
    os.write(_REAL_STDOUT, (json.dumps(obj) + "\n").encode())

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec (MI355X_MICROARCH.md, chip table)
HBM_COPY_CEILING_GBS = 6290.0  # measured float4-copy ceiling, same guide


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=10)
    p.add_argument("--warmup", type=int, default=3)
    p.add_argument("--level", type=int, default=None,
                   help="finest level L: grid 2^L.  Default 13 (8192^2, BASELINE's metric grid) on one GPU, "
                        "14 (16384^2, BASELINE config 4) on several")
    p.add_argument("--coarsest", type=int, default=7, help="coarsest level (PS:18)")
    p.add_argument("--mu1", type=int, default=10, help="pre-smoothing sweeps (PS:21)")
    p.add_argument("--mu2", type=int, default=10, help="post-smoothing sweeps (PS:22)")
    p.add_argument("--omega", type=float, default=2.0 / 3.0, help="Jacobi weight (PS:127)")
    p.add_argument("--smoother", choices=["jacobi", "rbgs"], default="jacobi")
    p.add_argument("--dtype", choices=["f64", "f32", "mixed"], default="f64")
    p.add_argument("--arith", choices=["fma", "separate"], default="fma",
                   help="rounding of the Jacobi update (mgx_config.arith): fma = v' = fma(c1, nb, fma(c0, v, c1 b)), "
                        "separate = the reference's five library calls as five roundings (PS:138-142)")
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--cpu-level", type=int, default=13, help="grid level of the bounded CPU-baseline sample")
    return p.parse_args()


DT = {"f32": 0, "f64": 1, "mixed": 2}
BYTES = {"f32": 4, "f64": 8, "mixed": 4}


def host_threads(omp_max):
    """Threads this process may really use: CPU affinity and cgroup quota, not
    the machine's core count (a 1-GPU box grants a 16-CPU share)."""
    n = omp_max
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except Exception:
        pass
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    env = os.environ.get("MGX_CPU_THREADS")
    if env:
        return max(1, int(env))
    return max(1, min(n, 16 * max(1, int(os.environ.get("WORLD_SIZE", "1")))))


def cpu_baseline(args):
    """The oracle's Jacobi sweep timed on this box's host cores (bounded sample).
    primary: matrix-free stencil, OpenMP over rows, all cores ('port');
    extra  : reference-shaped CSR SpMV + scal/scal/add/add (PS:137-145), 1 thread."""
    import numpy as np

    from oracle import pyoracle as po

    L = args.cpu_level
    n = (1 << L) - 1
    dt = np.float32 if args.dtype == "f32" else np.float64
    b = po.rhs_sine(L).astype(dt)
    u0 = po.fill_uniform((n, n), 12345).astype(dt)
    threads = host_threads(po.lib().orc_max_threads())
    # OpenMP flavour on the full-size grid (three 0.5 GB arrays: beyond the L3):
    # warm once, then ~10-30 core-seconds of sweeps
    po.baseline_jacobi("omp", u0, b, 1, args.omega, threads)
    sweeps = 250
    t_omp, _ = po.baseline_jacobi("omp", u0, b, sweeps, args.omega, threads)
    v_omp = n * n * sweeps / t_omp
    # reference-shaped flavour, one thread, on a quarter-size grid to stay bounded
    Lc = max(L - 1, 6)
    nc = (1 << Lc) - 1
    bc = np.ascontiguousarray(b[:nc, :nc])
    uc = np.ascontiguousarray(u0[:nc, :nc])
    sweeps_csr = 100
    t_csr, _ = po.baseline_jacobi("csr", uc, bc, sweeps_csr, args.omega)
    v_csr = nc * nc * sweeps_csr / t_csr
    model = "unknown"
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                model = line.split(":", 1)[1].strip()
                break
    except Exception:
        pass
    host = {"cpu_model": model, "nproc": os.cpu_count(), "threads_usable": threads}
    main = {"value": v_omp, "unit": "updates/s", "cores": threads, "kind": "port", "host": host,
            "sample": f"{sweeps} weighted-Jacobi sweeps, {n}^2 {args.dtype if args.dtype != 'mixed' else 'f64'} grid, "
                      f"matrix-free OpenMP oracle ({t_omp:.2f} s)"}
    extra = {"value": v_csr, "unit": "updates/s", "cores": 1, "kind": "port",
             "sample": f"{sweeps_csr} sweeps, {nc}^2, reference-shaped CSR SpMV + scal/scal/add/add + copy as PS:137-145 ({t_csr:.2f} s)"}
    return main, extra


def pmc_traffic(workload_key):
    """HBM bytes per launch of the dominant kernel, from the committed rocprofv3
    PMC passes (profiles/pmc_traffic.json, numbers from the tables tools/prof_summary.py
    prints); None if there is no measurement for this exact workload."""
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    try:
        with open(path) as fh:
            return json.load(fh).get(workload_key, {}).get("hbm_bytes_per_launch")
    except Exception:
        return None


def run_single(args):
    import __graft_entry__ as ge

    pkg = ge.load_package()
    L = args.level
    n = (1 << L) - 1
    cfg = dict(finest_level=L, coarsest_level=min(args.coarsest, L), mu0=0, mu1=args.mu1, mu2=args.mu2,
               omega=args.omega, smoother=1 if args.smoother == "rbgs" else 0, dtype=DT[args.dtype],
               schedule=pkg.SCHEDULE_V, profile=2, arith=pkg.ARITH_FMA if args.arith == "fma" else pkg.ARITH_SEPARATE)
    # profile = 2: the finest level's passes are launched one by one between HIP events, the rest of the
    # cycle is one hipGraph replay between two events (include/mgx.h) - mgx_solve's own execution, timed
    mg = pkg.Multigrid(**cfg)
    # synthetic input, generated on the device: resident in HBM before any timing
    mg.fill_rhs(1, 0.0)               # b = h^2 8 pi^2 sin(2 pi x) sin(2 pi y)
    mg.fill_guess_random(12345)       # u0 ~ U(-1,1)
    # ---- the "V-cycles to 1e-8" half of the metric (untimed here) ----
    mg.solve(tol=0.0, max_cycles=1)   # (graph capture, code objects, clocks: not part of a solve's own time)
    mg.fill_guess_random(12345)
    st0, hist0 = mg.solve(tol=1e-8, max_cycles=60)
    cycles_to_tol = st0.cycles if st0.converged else None
    solve_seconds = st0.seconds
    # the same cycles on an iterate that is still far from converged: the first five from a fresh random guess
    # (the K timed steps below mostly run on an iterate already at the rounding floor; same kernels, same bytes)
    mg.fill_guess_random(54321)
    mg.synchronize()
    t0 = time.perf_counter()
    st5, _ = mg.solve(tol=0.0, max_cycles=5)
    mg.synchronize()
    first5_ms = (time.perf_counter() - t0) / 5 * 1e3
    # ---- warmup + timed steps on a fresh random guess ----
    mg.fill_guess_random(12345)
    if args.warmup > 0:
        mg.solve(tol=0.0, max_cycles=args.warmup)
    mg.profile_reset()
    mg.synchronize()
    t0 = time.perf_counter()
    st, hist = mg.solve(tol=0.0, max_cycles=args.steps)
    mg.synchronize()
    t1 = time.perf_counter()
    prof = mg.profile()
    secs = t1 - t0
    assert st.cycles == args.steps
    updates = st.fine_updates
    value = updates / secs
    # ---- roofline of the dominant kernel: the finest-level smoother passes ----
    # `achieved` charges a launch with the bytes it MUST move, once: one pass reads v and b and
    # writes v' (3*sizeof(T)*n^2, SURVEY §8d's 12 / 24 B per point) however many sweeps it performs,
    # plus the folded transfers' coarse traffic (the restricted right-hand side written by the last
    # pre-smoothing pass, the coarse correction read by the first post-smoothing pass: sizeof(T)*n_c^2
    # each, once per V-cycle).  The K sweeps a fused pass performs per byte are reported separately
    # (`sweeps_per_pass`, `effective_gbs_per_sweep`): that figure is a throughput multiple, not a
    # roofline fraction.
    es = BYTES[args.dtype]
    sm_ms = prof["ms"][0]               # MGX_PROF_SMOOTH_FINE
    sm_launches = max(prof["launches"][0], 1)
    sm_sweeps = prof["sweeps"][0]
    avg_ms = sm_ms / sm_launches
    sweeps_per_launch = sm_sweeps / sm_launches
    nc = (1 << (L - 1)) - 1
    folded = sweeps_per_launch > 1.0 and L > cfg["coarsest_level"]
    pass_bytes = 3.0 * es * n * n
    coarse_bytes_per_cycle = (2.0 * es * nc * nc) if folded else 0.0
    alg_bytes_per_launch = pass_bytes + coarse_bytes_per_cycle * args.steps / sm_launches
    achieved = alg_bytes_per_launch / (avg_ms * 1e-3) / 1e9
    tname = "double" if es == 8 else "float"
    if args.smoother == "rbgs" and sweeps_per_launch <= 1.0:
        kernel = f"k_rbgs<{tname}>"
    elif sweeps_per_launch > 1.0:
        kernel = (f"k_jacobi_fused<{tname},K> / k_jacobi_cycle<{tname},K,PRE,POST> "
                  f"(finest-level smoother passes, K = {sweeps_per_launch:g} sweeps per pass on average; "
                  f"the cycle's correction, residual+restriction and norm ride in the same passes)")
    else:
        kernel = f"k_jacobi_rows<{tname}>"
    wl_key = f"L{L}_{args.smoother}_{args.dtype}_mu{args.mu1}" + ("_fma" if args.arith == "fma" and args.smoother == "jacobi" else "")
    traffic = pmc_traffic(wl_key)
    out = {
        "metric": "fine_grid_stencil_updates_per_sec",
        "value": value,
        "unit": "updates/s",
        "n_gpus": 1,
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
                        f"{args.dtype}, Jacobi update {'contracted (fma)' if args.arith == 'fma' else 'separately rounded'}, "
                        f"exact bottom solve at {(1 << cfg['coarsest_level']) - 1}^2, "
                        f"rhs h^2*8pi^2 sin(2pi x)sin(2pi y), u0~U(-1,1)",
            "finest_level": L, "coarsest_level": cfg["coarsest_level"], "mu1": args.mu1, "mu2": args.mu2,
            "smoother": args.smoother, "arith": args.arith, "step": "one V-cycle + residual norm (mgx_solve loop body)",
            "parallelism": "1 GPU",
        },
        "vcycles_to_1e-8": cycles_to_tol,
        "seconds_to_1e-8": solve_seconds,
        "ms_per_step_first_5_cycles_from_a_random_guess": first5_ms,      # incl. the initial residual norm (one more pass)
        "runtime_libs": pkg.runtime_libs(),
        "residual_history_to_1e-8": [float(x) for x in hist0],
        "roofline": {
            "bound": "hbm",
            "kernel": kernel,
            "achieved": achieved,
            "peak": HBM_PEAK_GBS,
            "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBS,
            "frac_of_measured_copy_ceiling": achieved / HBM_COPY_CEILING_GBS,
            "traffic": traffic,
            "traffic_source": ("profiles/pmc_traffic.json: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command "
                               "(committed summaries profiles/r0*_pmc_summary.md, written by tools/prof_summary.py), gfx950 "
                               "correction (2*FETCH_SIZE + WRITE_SIZE)*1024; NOT measured in this run") if traffic else None,
            "hbm_physical_gbs": (traffic / (avg_ms * 1e-3) / 1e9) if traffic else None,
            "algorithmic_bytes_per_launch": alg_bytes_per_launch,
            "sweeps_per_pass": sweeps_per_launch,
            "effective_gbs_per_sweep": 3.0 * es * n * n * sweeps_per_launch / (avg_ms * 1e-3) / 1e9,
            "avg_launch_ms": avg_ms,
            "launches_timed": sm_launches,
            "sweeps_timed": sm_sweeps,
            "smoother_updates_per_s": n * n * sm_sweeps / (sm_ms * 1e-3),
            "how": "HIP events on the solver's stream around every finest-level smoothing block inside the timed steps "
                   "(cfg.profile = 2: those passes launched one by one, the levels below replayed from one hipGraph)",
        },
        "phase_ms_per_step": {
            "smooth_fine": prof["ms"][0] / args.steps, "restrict_fine": prof["ms"][1] / args.steps,
            "prolong_fine": prof["ms"][2] / args.steps, "norm_fine": prof["ms"][3] / args.steps,
            "coarse_levels": prof["ms"][4] / args.steps,
        },
    }
    mg.close()
    # ---- the same K steps the way mgx_solve runs them when nobody is timing phases (cfg.profile = 0: the whole cycle and its
    # norm replayed from ONE hipGraph, no events, no eager launches).  Informational: `value` above stays the number of the
    # instrumented region the roofline figures come from.
    try:
        with pkg.Multigrid(**dict(cfg, profile=0)) as mg0:
            mg0.fill_rhs(1, 0.0)
            mg0.fill_guess_random(12345)
            mg0.solve(tol=0.0, max_cycles=max(args.warmup, 2))
            mg0.solve(tol=0.0, max_cycles=args.steps)           # (every graph of the loop captured before anything is timed)
            t_off = []
            for k_off in (args.steps, 2 * args.steps):          # the difference drops what a solve call costs once (its initial norm)
                mg0.synchronize()
                t0 = time.perf_counter()
                st_off, _ = mg0.solve(tol=0.0, max_cycles=k_off)
                mg0.synchronize()
                t_off.append(time.perf_counter() - t0)
            s_off = t_off[1] - t_off[0]
        out["profiling_off"] = {
            "ms_per_step": s_off / args.steps * 1e3, "value": st_off.fine_updates / 2.0 / s_off, "unit": "updates/s",
            "how": f"a second handle with cfg.profile = 0 (one hipGraph replay per cycle + norm): wall time of {2 * args.steps} steps minus "
                   f"that of {args.steps} steps",
        }
    except Exception as e:
        out["profiling_off"] = {"error": str(e)}
    # ---- the HBM-bound regime, live: the same smoother with temporal fusion off
    # (one k_jacobi_rows / k_rbgs launch per sweep moves exactly its algorithmic
    # bytes), timed with HIP events on the solver's stream.  The fused passes
    # above beat the roofline by reusing bytes; this is the roofline they start from.
    try:
        os.environ["MGX_FUSE"] = "1"
        with pkg.Multigrid(**dict(cfg, profile=0)) as mg1:
            mg1.fill_rhs(1, 0.0)
            mg1.fill_guess_random(12345)
            mg1.time_smoother(5)
            sweeps1 = 30
            ms1 = mg1.time_smoother(sweeps1)
        a1 = 3.0 * es * n * n / (ms1 / sweeps1 * 1e-3) / 1e9
        out["roofline_single_sweep"] = {
            "bound": "hbm", "kernel": ("k_rbgs" if args.smoother == "rbgs" else "k_jacobi_rows") + f"<{tname}>",
            "achieved": a1, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": a1 / HBM_PEAK_GBS,
            "frac_of_measured_copy_ceiling": a1 / HBM_COPY_CEILING_GBS,
            "traffic": pmc_traffic(f"L{L}_{args.smoother}_{args.dtype}_mu1_unfused"),
            "avg_launch_ms": ms1 / sweeps1, "launches_timed": sweeps1,
            "how": "MGX_FUSE=1 handle, mgx_time_smoother: HIP events around 30 back-to-back sweeps",
        }
    except Exception as e:
        out["roofline_single_sweep"] = {"error": str(e)}
    finally:
        os.environ.pop("MGX_FUSE", None)
    # ---- the reference's own problem and schedule (PS:123 f = 4, PS:630 zero guess,
    # PS:727 fullmultigrid), run to 1e-8: one FMG pass (mu0 = 0: one V-cycle per level)
    # followed by V-cycles.  Untimed part of the job; reported for the "V-cycles to 1e-8"
    # half of the metric.
    try:
        cfg2 = dict(cfg, schedule=pkg.SCHEDULE_FMG, mu0=0, profile=0)
        with pkg.Multigrid(**cfg2) as mg2:
            mg2.fill_rhs(0, 4.0)
            mg2.solve(tol=1e-8, max_cycles=60)          # warm (kernels loaded, clocks up)
            mg2.zero_level(L, 0)
            st2, hist2 = mg2.solve(tol=1e-8, max_cycles=60)
            out["reference_problem_fmg"] = {
                "problem": "-Laplace u = 4, u = 0 on the boundary (PS:123, 283-335), zero guess, fullmultigrid (PS:629-650) "
                           "with one V-cycle per level, then V-cycles",
                "cycles_to_1e-8": st2.cycles if st2.converged else None, "seconds_to_1e-8": st2.seconds,
                "residual_history": [float(x) for x in hist2],
            }
    except Exception as e:                                 # never lose the headline line
        out["reference_problem_fmg"] = {"error": str(e)}
    if not args.no_cpu_baseline:
        main, extra = cpu_baseline(args)
        out["cpu_baseline"] = main
        out["cpu_baseline_reference_shaped"] = extra
    emit(out)


def free_port():
    import socket

    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        return so.getsockname()[1]


def visible_devices():
    """HIP devices this process could use, counted WITHOUT any runtime (the launcher must stay a process that
    never touched the GPU: its children are the ranks): the visibility variables if set, else the KFD
    topology nodes that have compute units."""
    for var in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v is not None:
            return len([x for x in v.split(",") if x.strip() != ""])
    n = 0
    base = "/sys/class/kfd/kfd/topology/nodes"
    try:
        for node in os.listdir(base):
            try:
                props = dict(ln.split()[:2] for ln in open(os.path.join(base, node, "properties")) if len(ln.split()) >= 2)
                if int(props.get("simd_count", "0")) > 0:
                    n += 1
            except (OSError, ValueError):
                pass
    except OSError:
        return 0
    return n


def null_line(args, n, error):
    return {"metric": "fine_grid_stencil_updates_per_sec", "value": None, "unit": "updates/s", "n_gpus": n,
            "steps": args.steps, "warmup": args.warmup, "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": f"2D Poisson {1 << (args.level or 14)}^2, row slabs over {n} GPUs"}, "error": error}


def launch_ranks(args, argv):
    """`python bench.py --gpus N` started as ONE plain process (no torchrun): become the launcher.
    Start exactly N child ranks of this same script, one per GPU, with RANK / LOCAL_RANK /
    WORLD_SIZE / MASTER_* set; relay rank 0's single JSON line.  Every child is supervised: on the
    first non-zero exit (a GPU fault, an RCCL error, a timed-out rendezvous) the others are terminated,
    then killed, a `value: null` line is printed and the launcher exits non-zero; the whole job has a
    deadline (MGX_BENCH_DEADLINE seconds, default 1500).  Nothing here touches the GPU and nothing is
    exec'ed over a process that did."""
    import subprocess
    import tempfile

    n = args.gpus
    rehearsal = bool(os.environ.get("MGX_DIST_SINGLE_DEVICE")) or bool(os.environ.get("MGX_BENCH_DRYRUN"))
    ndev = visible_devices()
    if ndev < n and not rehearsal:
        emit(null_line(args, n, f"--gpus {n} needs {n} HIP devices, {ndev} visible (rehearsal on fewer devices: "
                                f"MGX_DIST_SINGLE_DEVICE=1 MGX_DIST_BACKEND=staged)"))
        sys.exit(2)
    port, rdzv_port = free_port(), free_port()
    procs = []
    out0 = tempfile.TemporaryFile()
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), MGX_RDZV_PORT=str(rdzv_port), MGX_BENCH_LAUNCHED="1")
        # rank 0's stdout is the result line; every rank's stderr goes to ours
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env,
                                      stdout=out0 if r == 0 else subprocess.DEVNULL, stderr=2))
    if os.environ.get("MGX_BENCH_LAUNCH_LOG"):       # tests: which ranks were started
        with open(os.environ["MGX_BENCH_LAUNCH_LOG"], "w") as fh:
            json.dump({"ranks": n, "pids": [p.pid for p in procs], "port": port}, fh)
    deadline = time.monotonic() + float(os.environ.get("MGX_BENCH_DEADLINE", "1500"))
    failure = None
    while failure is None:
        codes = [p.poll() for p in procs]
        bad = [(r, c) for r, c in enumerate(codes) if c not in (None, 0)]
        if bad:
            failure = "rank(s) " + ", ".join(f"{r} exited with code {c}" for r, c in bad)
        elif all(c == 0 for c in codes):
            break
        elif time.monotonic() > deadline:
            failure = "deadline exceeded"
        else:
            time.sleep(0.05)
    if failure is not None:
        for p in procs:                              # exactly the processes started here, by pid
            if p.poll() is None:
                p.terminate()
        t_end = time.monotonic() + 5.0
        for p in procs:
            try:
                p.wait(timeout=max(0.1, t_end - time.monotonic()))
            except subprocess.TimeoutExpired:
                p.kill()
                p.wait()
    out0.seek(0)
    line = None
    for ln in out0.read().decode(errors="replace").splitlines():
        if ln.lstrip().startswith("{"):
            line = ln
    codes = [p.returncode for p in procs]
    if failure is None and line is not None:
        os.write(_REAL_STDOUT, (line + "\n").encode())
        return
    sys.stderr.write(f"bench.py launcher: {failure or 'no result line'}; rank exit codes {codes}\n")
    # a rank's own error line (value null) is relayed; anything else is replaced by one
    relay = None
    if line is not None:
        try:
            relay = line if json.loads(line).get("value") is None else None
        except ValueError:
            relay = None
    if relay is not None:
        os.write(_REAL_STDOUT, (relay + "\n").encode())
    else:
        emit(null_line(args, n, f"launcher: {failure or 'rank 0 printed no result line'}; rank exit codes {codes}"))
    sys.exit(1)


def dry_run(args, world):
    """MGX_BENCH_DRYRUN=1 (tests/test_bench_launcher.py, CPU): every rank joins the job's store and is
    counted by a collective; rank 0 prints the line shape with the rank count the collective saw.
    No solver, no GPU: this exercises the launcher and the rendezvous only."""
    seen = 1
    rank = int(os.environ.get("RANK", "0"))
    if os.environ.get("MGX_BENCH_FAIL_RANK") == str(rank):
        sys.exit(7)                     # test hook: a rank that dies before the rendezvous
    if world > 1:
        import __graft_entry__ as ge

        ge.load_package()
        from multigrid_nikhil_c_amd.rendezvous import Store

        store = Store(rank, world, timeout=float(os.environ.get("MGX_RDZV_TIMEOUT", "20")))
        seen = store.allreduce_sum_int(1)
        if os.environ.get("MGX_BENCH_FAIL_LATE_RANK") == str(rank):
            os._exit(9)                 # test hook: a rank that dies in the middle of the job
        if os.environ.get("MGX_BENCH_HANG_RANK") == str(rank):
            time.sleep(3600)            # test hook: a rank that never comes back
        store.barrier()
        store.close()
    if seen != args.gpus:
        sys.stderr.write(f"bench.py: --gpus {args.gpus} but the collective saw {seen} ranks\n")
        sys.exit(3)
    if rank == 0:
        emit({"metric": "fine_grid_stencil_updates_per_sec", "value": None, "unit": "updates/s", "n_gpus": seen,
              "steps": args.steps, "warmup": args.warmup, "dry_run": True, "ranks_seen_by_collective": seen})


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        launch_ranks(args, sys.argv[1:])
        return
    if args.gpus != world and not (args.gpus <= 1 and world <= 1):
        # never report a GPU count other than the ranks that really run
        sys.stderr.write(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}; refusing to run\n")
        sys.exit(2)
    if os.environ.get("MGX_BENCH_DRYRUN"):
        dry_run(args, world)
        return
    if args.gpus <= 1 and world <= 1:
        if args.level is None:
            args.level = 13
        run_single(args)
        return
    if args.level is None:
        args.level = 14
    from importlib import import_module

    import __graft_entry__ as ge

    ge.load_package()
    dist_bench = import_module("multigrid_nikhil_c_amd.dist_bench")
    try:
        dist_bench.run(args, emit)
    except BaseException as e:       # noqa: BLE001 - report, then fail: the line is the only channel back
        import traceback

        traceback.print_exc()
        if int(os.environ.get("RANK", "0")) == 0:
            emit(null_line(args, world, f"{type(e).__name__}: {e}"))
        raise


if __name__ == "__main__":
    main()
