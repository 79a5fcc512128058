#!/bin/bash
mkdir -p gpurun_out/r02
timeout -k 10 600 python -m pytest tests/test_gpu_api.py -x -q -m gpu > gpurun_out/r02/z_tests.log 2>&1 || { tail -40 gpurun_out/r02/z_tests.log; exit 1; }
tail -3 gpurun_out/r02/z_tests.log
python bench.py --steps 20 --warmup 3 > gpurun_out/r02/bench_z.json 2> gpurun_out/r02/bench_z.err || { tail -5 gpurun_out/r02/bench_z.err; exit 1; }
python - <<'PY'
import json
d=json.load(open('gpurun_out/r02/bench_z.json'))
print(d['ms_per_step'], d['value'], d['phase_ms_per_step'])
print(d['roofline'])
print(d['cpu_baseline'])
PY
