set -e
mkdir -p gpurun_out/r02
python -m pytest tests/test_gpu_solve.py -x -q -k "deep_folded or knobs" > gpurun_out/r02/pytest_e1.log 2>&1 || { tail -40 gpurun_out/r02/pytest_e1.log; exit 1; }
tail -2 gpurun_out/r02/pytest_e1.log
run() { tag=$1; shift; env "$@" python bench.py --no-cpu-baseline > gpurun_out/r02/bench_e_$tag.json 2>/dev/null; python - "gpurun_out/r02/bench_e_$tag.json" "$tag" <<'PY'
import json,sys
d=json.load(open(sys.argv[1]))
print(sys.argv[2], round(d["ms_per_step"],3), round(d["value"]/1e9,1), {k:round(v,3) for k,v in d["phase_ms_per_step"].items()}, round(d["roofline"]["frac"],3), round(d["roofline"]["avg_launch_ms"],4), d["roofline"]["launches_timed"])
PY
}
run prev10 MGX_LIBMGX_PATH=$PWD/tools/ab/libmgx_prev.so MGX_PLAN_PRE=10 MGX_PLAN_POST=10
run new10 MGX_PLAN_PRE=10 MGX_PLAN_POST=10
run prev10b MGX_LIBMGX_PATH=$PWD/tools/ab/libmgx_prev.so MGX_PLAN_PRE=10 MGX_PLAN_POST=10
run new10b MGX_PLAN_PRE=10 MGX_PLAN_POST=10
run new10_4096 MGX_PLAN_PRE=10 MGX_PLAN_POST=10 MGX_PLAN_MIN_N=4096
run new10_2048 MGX_PLAN_PRE=10 MGX_PLAN_POST=10 MGX_PLAN_MIN_N=2048
run new_default A=1
run new_8_2 MGX_PLAN_PRE=2,8 MGX_PLAN_POST=8,2
