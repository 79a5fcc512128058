#!/bin/bash
timeout -k 10 600 python -m pytest tests/test_gpu_operators.py tests/test_gpu_api.py tests/test_gpu_var.py -x -q -m gpu 2>&1 | tail -2
b() { python bench.py --no-cpu-baseline "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print(round(d['ms_per_step'],4), {k:round(v,4) for k,v in d['phase_ms_per_step'].items() if k in ('smooth_fine','coarse_levels')}, round(d['roofline']['avg_launch_ms'],4), round(d['roofline']['frac'],3))"; }
echo -n "L10 (reference hierarchy): "; b --level 10 --steps 50 --warmup 5
echo -n "L10 again: "; b --level 10 --steps 50 --warmup 5
echo -n "L13: "; b --level 13 --steps 20 --warmup 3
echo -n "L13: "; b --level 13 --steps 20 --warmup 3
