"""bench.py --gpus N started as ONE plain process (the way the driver invokes it) must create its
own N ranks.  CPU test: MGX_BENCH_DRYRUN=1 replaces the solver by a gloo all_reduce that counts
the ranks, so what is exercised is the launcher, the environment it gives each rank and the
rendezvous - the parts that were missing in round 1 (WORLD_SIZE unset -> one rank, n_gpus 1)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_bench(n, tmp_path, extra_env=None, args=()):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    log = tmp_path / "launch.json"
    env.update({"MGX_BENCH_DRYRUN": "1", "MGX_BENCH_LAUNCH_LOG": str(log)})
    env.update(extra_env or {})
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--steps", "2", "--warmup", "1", *args],
                       env=env, capture_output=True, text=True, timeout=300)
    return r, log


@pytest.mark.parametrize("n", [2, 4])
def test_plain_invocation_starts_exactly_n_ranks(tmp_path, n):
    r, log = run_bench(n, tmp_path)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, lines                         # the contract: ONE JSON line
    out = json.loads(lines[0])
    assert out["n_gpus"] == n and out["ranks_seen_by_collective"] == n
    started = json.loads(log.read_text())
    assert started["ranks"] == n and len(set(started["pids"])) == n


def test_a_failing_rank_fails_the_launcher(tmp_path):
    # rank 1 dies before the rendezvous: no line may be taken as a result and the exit code is non-zero
    r, _ = run_bench(2, tmp_path, extra_env={"MGX_BENCH_DRYRUN": "1", "MGX_BENCH_FAIL_RANK": "1"})
    assert r.returncode != 0


def test_world_size_mismatch_is_refused(tmp_path):
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", MGX_BENCH_DRYRUN="1")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4"], env=env, capture_output=True,
                       text=True, timeout=120)
    assert r.returncode != 0 and "refusing" in r.stderr


def test_too_few_devices_is_an_error_line_not_a_silent_single_rank_run(tmp_path):
    # no rehearsal environment, no GPUs in this container: the launcher must say so, not run 1 rank
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "MGX_DIST_SINGLE_DEVICE", "MGX_BENCH_DRYRUN")}
    import torch

    if torch.cuda.device_count() >= 2:
        pytest.skip("this box has the devices")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], env=env, capture_output=True,
                       text=True, timeout=120)
    assert r.returncode != 0
    out = json.loads(r.stdout.strip().splitlines()[-1])
    assert out["value"] is None and out["n_gpus"] == 2 and "needs 2 HIP devices" in out["error"]
