#!/bin/bash
timeout -k 10 1000 python -m pytest tests/test_gpu_slabs.py tests/test_gpu_dist.py tests/test_gpu_operators.py tests/test_gpu_solve.py -x -q -m gpu 2>&1 | tail -3
python tools/slab_budget.py 2>&1 | tail -2
python tools/slab_budget.py 2>&1 | tail -2
b() { python bench.py --no-cpu-baseline --level $1 --steps 30 --warmup 3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print(round(d['ms_per_step'],4), {k:round(v,4) for k,v in d['phase_ms_per_step'].items() if k in ('smooth_fine','coarse_levels')}, round(d['roofline']['avg_launch_ms'],4), round(d['roofline']['frac'],3))"; }
for L in 13 12 13 12; do echo -n "L$L: "; b $L; done
