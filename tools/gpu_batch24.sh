#!/bin/bash
export MGX_PAIR_RATIO=130 MGX_PAIR_MIN_ROWS=200
b() { python bench.py --no-cpu-baseline --level $1 --steps 30 --warmup 3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print(round(d['ms_per_step'],4), {k:round(v,4) for k,v in d['phase_ms_per_step'].items() if k in ('smooth_fine','coarse_levels')}, round(d['roofline']['avg_launch_ms'],4), round(d['roofline']['frac'],3))"; }
for r in 1 2; do
echo -n "L13 pair off: "; MGX_PAIR=0 b 13
echo -n "L13 edge 23 last 38: "; b 13
echo -n "L13 edge 10 last 38: "; MGX_EDGE_PCT=10 b 13
echo -n "L13 edge 23 last 25: "; MGX_LAST_PCT=25 b 13
echo -n "L13 edge 10 last 25: "; MGX_EDGE_PCT=10 MGX_LAST_PCT=25 b 13
echo -n "L13 edge 30 last 45: "; MGX_EDGE_PCT=30 MGX_LAST_PCT=45 b 13
echo -n "L13 edge  0 last  0: "; MGX_EDGE_PCT=0 MGX_LAST_PCT=0 b 13
done
