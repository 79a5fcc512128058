set -e
mkdir -p gpurun_out/r02
python -m pytest tests -m gpu -x -q > gpurun_out/r02/pytest_g.log 2>&1 || { tail -40 gpurun_out/r02/pytest_g.log; exit 1; }
tail -2 gpurun_out/r02/pytest_g.log
python bench.py > gpurun_out/r02/bench_g.json 2> gpurun_out/r02/bench_g.err
python - <<'PY'
import json
d=json.load(open("gpurun_out/r02/bench_g.json"))
print("default", round(d["ms_per_step"],3), round(d["value"]/1e9,1), {k:round(v,3) for k,v in d["phase_ms_per_step"].items()}, d["roofline"]["frac"], d["roofline"]["avg_launch_ms"], d["roofline"]["launches_timed"], d["roofline_single_sweep"].get("frac"), d["cpu_baseline"]["value"])
PY
python tools/run_configs.py > gpurun_out/r02/configs_g.md 2> gpurun_out/r02/configs_g.err
cat gpurun_out/r02/configs_g.md
