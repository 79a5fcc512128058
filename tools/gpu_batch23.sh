#!/bin/bash
b() { python bench.py --no-cpu-baseline --level $1 --steps 30 --warmup 3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print(round(d['ms_per_step'],4), {k:round(v,4) for k,v in d['phase_ms_per_step'].items() if k in ('smooth_fine','coarse_levels')}, round(d['roofline']['avg_launch_ms'],4), round(d['roofline']['frac'],3))"; }
for r in 1 2 3; do
echo -n "L13 pair off: "; MGX_PAIR=0 b 13
for ratio in 100 110 120 130 140; do echo -n "L13 pair ratio $ratio: "; MGX_PAIR_RATIO=$ratio b 13; done
done
for r in 1 2; do
echo -n "L12 pair off: "; MGX_PAIR=0 b 12
for ratio in 100 115 130; do echo -n "L12 pair ratio $ratio: "; MGX_PAIR_RATIO=$ratio b 12; done
done
echo "8-slab budget"; MGX_PAIR=0 python tools/slab_budget.py 2>&1 | tail -1; for ratio in 100 115 130; do echo -n "ratio $ratio: "; MGX_PAIR_RATIO=$ratio python tools/slab_budget.py 2>&1 | tail -1; done;  MGX_PAIR=0 python tools/slab_budget.py 2>&1 | tail -1
