#!/bin/bash
b() { python bench.py --no-cpu-baseline --level $1 --steps 10 --warmup 2 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print(round(d['ms_per_step'],4), {k:round(v,4) for k,v in d['phase_ms_per_step'].items() if k in ('smooth_fine','coarse_levels')}, round(d['roofline']['avg_launch_ms'],4), round(d['roofline']['frac'],3))"; }
for r in 1 2; do
for R in 0 252 300 380 480; do echo -n "L14 MGX_FUSE_ROWS=$R: "; MGX_FUSE_ROWS=$R b 14; done
done
