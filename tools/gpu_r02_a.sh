set -e
mkdir -p gpurun_out/r02
python -m pytest tests -m gpu -x -q > gpurun_out/r02/pytest_a.log 2>&1 || { tail -30 gpurun_out/r02/pytest_a.log; exit 1; }
tail -3 gpurun_out/r02/pytest_a.log
python bench.py --no-cpu-baseline > gpurun_out/r02/bench_a.json 2> gpurun_out/r02/bench_a.err
MGX_FUSE_ROWS=72 python bench.py --no-cpu-baseline > gpurun_out/r02/bench_a_r72.json 2>/dev/null
MGX_FUSE_ROWS=144 python bench.py --no-cpu-baseline > gpurun_out/r02/bench_a_r144.json 2>/dev/null
MGX_PLAN_PRE=10 MGX_PLAN_POST=10 python bench.py --no-cpu-baseline > gpurun_out/r02/bench_a_p10.json 2>/dev/null
MGX_PLAN_PRE=10 MGX_PLAN_POST=10 MGX_PLAN_MIN_N=4096 python bench.py --no-cpu-baseline > gpurun_out/r02/bench_a_p10b.json 2>/dev/null
MGX_PLAN_PRE=8,2 MGX_PLAN_POST=8,2 python bench.py --no-cpu-baseline > gpurun_out/r02/bench_a_p82.json 2>/dev/null
for f in gpurun_out/r02/bench_a*.json; do python - "$f" <<'PY'
import json,sys
d=json.load(open(sys.argv[1]))
print(sys.argv[1], round(d["ms_per_step"],3), round(d["value"]/1e9,1), {k:round(v,3) for k,v in d["phase_ms_per_step"].items()}, round(d["roofline"]["frac"],3), round(d["roofline"]["avg_launch_ms"],4), d["roofline"]["launches_timed"])
PY
done
