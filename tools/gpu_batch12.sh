#!/bin/bash
echo "== bit check with the two-height one-round geometry"
MGX_TWO_CLASS=148 timeout -k 10 600 python -m pytest tests/test_gpu_fma.py tests/test_gpu_dist.py -x -q -m gpu 2>&1 | tail -2
b() { python bench.py --no-cpu-baseline --steps 20 --warmup 3 "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print(round(d['ms_per_step'],4), {k:round(v,4) for k,v in d['phase_ms_per_step'].items() if k in ('smooth_fine','coarse_levels')}, round(d['roofline']['avg_launch_ms'],4), round(d['roofline']['frac'],3))"; }
echo "== MGX_TWO_CLASS = 100 x tall / short (0 = the default two-round geometry)"
for L in 13 12 14; do for t in 0 100 125 148 170 200 0; do echo -n "L$L two_class $t: "; MGX_TWO_CLASS=$t b --level $L; done; done
echo "== 8-slab budget"
for t in 0 148 170; do echo "two_class $t"; MGX_TWO_CLASS=$t python tools/slab_budget.py 14 fma 2>&1 | grep "P=8"; done
