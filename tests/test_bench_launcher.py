"""bench.py --gpus N started as ONE plain process (the way the driver invokes it) must create its
own N ranks.  CPU test: MGX_BENCH_DRYRUN=1 replaces the solver by a collective of the job's TCP store
(rendezvous.py) that counts the ranks, so what is exercised is the launcher, the environment it gives
each rank, the rendezvous and the supervision of the children (a rank that dies or hangs must end
the job with a `value: null` line and a non-zero exit code, promptly)."""
import time

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
    sys.path.insert(0, ROOT)
    import bench

    if bench.visible_devices() >= 2:
        pytest.skip("this box has the devices")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], env=env, capture_output=True,
                       text=True, timeout=120)
    assert r.returncode != 0
    out = json.loads(r.stdout.strip().splitlines()[-1])
    assert out["value"] is None and out["n_gpus"] == 2 and "needs 2 HIP devices" in out["error"]


def test_a_rank_dying_mid_job_ends_the_job_promptly_with_a_null_line(tmp_path):
    # rank 1 passes the rendezvous, then dies: rank 0 would wait in its next collective; the launcher must
    # terminate it, print value: null and exit non-zero without waiting for any collective's own timeout
    t0 = time.monotonic()
    r, _ = run_bench(2, tmp_path, extra_env={"MGX_BENCH_FAIL_LATE_RANK": "1", "MGX_RDZV_TIMEOUT": "120"})
    assert r.returncode != 0
    assert time.monotonic() - t0 < 30
    out = json.loads(r.stdout.strip().splitlines()[-1])
    assert out["value"] is None and out["n_gpus"] == 2 and "exited with code 9" in out["error"]


def test_a_hanging_rank_is_killed_at_the_deadline(tmp_path):
    t0 = time.monotonic()
    r, log = run_bench(2, tmp_path, extra_env={"MGX_BENCH_HANG_RANK": "1", "MGX_RDZV_TIMEOUT": "120", "MGX_BENCH_DEADLINE": "3"})
    assert r.returncode != 0 and time.monotonic() - t0 < 30
    out = json.loads(r.stdout.strip().splitlines()[-1])
    assert out["value"] is None and "deadline" in out["error"]
    for pid in json.loads(log.read_text())["pids"]:          # nobody is left behind
        with pytest.raises(ProcessLookupError):
            os.kill(pid, 0)


def test_ranks_under_a_foreign_launcher_find_each_other_through_the_rendezvous_file(tmp_path):
    """the driver starts the ranks with `python -m torch.distributed.run`: no MGX_RDZV_PORT, MASTER_PORT belongs
    to that launcher's own store - rank 0 publishes an ephemeral port in a file keyed by the launcher's pid"""
    env = {k: v for k, v in os.environ.items() if k not in ("MGX_RDZV_PORT",)}
    env.update(MGX_BENCH_DRYRUN="1", WORLD_SIZE="3", MASTER_ADDR="127.0.0.1", MASTER_PORT="29999")
    procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "3", "--steps", "1", "--warmup", "0"],
                              env=dict(env, RANK=str(r), LOCAL_RANK=str(r)), stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
             for r in range(3)]
    outs = [p.communicate(timeout=120) for p in procs]
    assert [p.returncode for p in procs] == [0, 0, 0], [o[1][-500:] for o in outs]
    line = json.loads(outs[0][0].strip().splitlines()[-1])
    assert line["ranks_seen_by_collective"] == 3 and not outs[1][0].strip() and not outs[2][0].strip()
