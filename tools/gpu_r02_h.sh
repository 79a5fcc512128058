set -e
mkdir -p gpurun_out/r02
python -m pytest tests -m gpu -x -q > gpurun_out/r02/pytest_h.log 2>&1 || { tail -40 gpurun_out/r02/pytest_h.log; exit 1; }
tail -2 gpurun_out/r02/pytest_h.log
python -c "import __graft_entry__ as g; g.smoke()"
python bench.py > gpurun_out/r02/bench_h.json 2> gpurun_out/r02/bench_h.err
python - <<'PY'
import json
d=json.load(open("gpurun_out/r02/bench_h.json"))
print("default", round(d["ms_per_step"],3), round(d["value"]/1e9,1), {k:round(v,3) for k,v in d["phase_ms_per_step"].items()}, round(d["roofline"]["frac"],3), d["roofline"]["avg_launch_ms"], d["roofline"]["launches_timed"])
PY
python bench.py --dtype mixed --mu1 2 --mu2 1 --no-cpu-baseline > gpurun_out/r02/bench_h_mixed.json 2>/dev/null
python bench.py --dtype f32 --no-cpu-baseline > gpurun_out/r02/bench_h_f32.json 2>/dev/null
python bench.py --smoother rbgs --mu1 2 --mu2 2 --no-cpu-baseline > gpurun_out/r02/bench_h_rbgs.json 2>/dev/null
for t in mixed f32 rbgs; do python -c "
import json; d=json.load(open('gpurun_out/r02/bench_h_$t.json')); print('$t', round(d['ms_per_step'],3), round(d['value']/1e9,1), d['vcycles_to_1e-8'])"; done
