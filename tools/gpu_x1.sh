#!/bin/bash
b() { python bench.py --no-cpu-baseline --steps 20 --warmup 3 "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print(round(d['ms_per_step'],4), {k:round(v,4) for k,v in d['phase_ms_per_step'].items() if v>0})"; }
for r in 1 2; do
for m in 150 100 64 32; do
echo -n "mixed V(10,10) pair_min $m: "; MGX_PAIR_MIN_ROWS=$m b --level 13 --dtype mixed
done
echo -n "mixed V(10,10) pair_min 64 ratio 115: "; MGX_PAIR_MIN_ROWS=64 MGX_PAIR_RATIO=115 b --level 13 --dtype mixed
echo -n "f32 pair_min 150: "; b --level 13 --dtype f32
echo -n "f32 pair_min 64: "; MGX_PAIR_MIN_ROWS=64 b --level 13 --dtype f32
done
