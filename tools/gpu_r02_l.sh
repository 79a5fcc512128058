set -e
mkdir -p gpurun_out/r02
python -m pytest tests/test_gpu_solve.py tests/test_gpu_operators.py tests/test_gpu_slabs.py -x -q > gpurun_out/r02/pytest_l.log 2>&1 || { tail -40 gpurun_out/r02/pytest_l.log; exit 1; }
tail -2 gpurun_out/r02/pytest_l.log
run() { tag=$1; shift; python bench.py --no-cpu-baseline "$@" > gpurun_out/r02/bench_l_$tag.json 2>/dev/null; python - "gpurun_out/r02/bench_l_$tag.json" "$tag" <<'PY'
import json,sys
d=json.load(open(sys.argv[1]))
print(sys.argv[2], round(d["ms_per_step"],4), round(d["value"]/1e9,1), {k:round(v,3) for k,v in d["phase_ms_per_step"].items() if k in ("smooth_fine","coarse_levels")}, round(d["roofline"]["avg_launch_ms"],4), d["roofline"]["launches_timed"], d["vcycles_to_1e-8"])
PY
}
for lib in prev new prev new; do
  if [ $lib = prev ]; then export MGX_LIBMGX_PATH=$PWD/tools/ab/libmgx_prev.so; else unset MGX_LIBMGX_PATH; fi
  run ${lib}_c2 --level 12 --coarsest 7 --mu1 2 --mu2 1
  run ${lib}_c3 --level 13 --smoother rbgs --mu1 2 --mu2 1
  run ${lib}_j21 --level 13 --mu1 2 --mu2 1
  run ${lib}_j43 --level 13 --mu1 4 --mu2 3
done
