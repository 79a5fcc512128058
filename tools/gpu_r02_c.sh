set -e
mkdir -p gpurun_out/r02
python -m pytest tests/test_gpu_solve.py -x -q -k "deep_folded or knobs or folded" > gpurun_out/r02/pytest_c1.log 2>&1 || { tail -40 gpurun_out/r02/pytest_c1.log; exit 1; }
tail -3 gpurun_out/r02/pytest_c1.log
for v in "" "MGX_PLAN_PRE=10 MGX_PLAN_POST=10" "MGX_PLAN_PRE=2,8 MGX_PLAN_POST=10" "MGX_PLAN_PRE=8,2 MGX_PLAN_POST=8,2" "MGX_PLAN_PRE=10 MGX_PLAN_POST=10 MGX_PLAN_MIN_N=2048" "MGX_PLAN_PRE=10 MGX_PLAN_POST=10 MGX_FUSE_ROWS=128" "MGX_PLAN_PRE=10 MGX_PLAN_POST=10 MGX_FUSE_ROWS=32"; do
  tag=$(echo "$v" | tr ' =,' '___'); [ -z "$tag" ] && tag=default
  env $v python bench.py --no-cpu-baseline > gpurun_out/r02/bench_c_$tag.json 2>/dev/null
  python - "gpurun_out/r02/bench_c_$tag.json" "$tag" <<'PY'
import json,sys
d=json.load(open(sys.argv[1]))
print(sys.argv[2], round(d["ms_per_step"],3), round(d["value"]/1e9,1), {k:round(v,3) for k,v in d["phase_ms_per_step"].items()}, round(d["roofline"]["frac"],3), round(d["roofline"]["avg_launch_ms"],4), d["roofline"]["launches_timed"], d["vcycles_to_1e-8"])
PY
done
python -m pytest tests -m gpu -x -q > gpurun_out/r02/pytest_c.log 2>&1 || { tail -40 gpurun_out/r02/pytest_c.log; exit 1; }
tail -3 gpurun_out/r02/pytest_c.log
