#!/bin/bash
timeout -k 10 900 python -m pytest tests/test_gpu_operators.py tests/test_gpu_fma.py tests/test_gpu_solve.py -x -q -m gpu 2>&1 | tail -3
b() { python bench.py --no-cpu-baseline --steps 20 --warmup 3 "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print(round(d['ms_per_step'],4), {k:round(v,4) for k,v in d['phase_ms_per_step'].items()})"; }
for r in 1 2; do
for a in 0 1; do
echo -n "mixed V(10,10) f32 auto rows $a: "; MGX_F32_AUTO_ROWS=$a b --level 13 --dtype mixed
echo -n "f32 V(10,10) auto rows $a: "; MGX_F32_AUTO_ROWS=$a b --level 13 --dtype f32
echo -n "mixed L12 auto rows $a: "; MGX_F32_AUTO_ROWS=$a b --level 12 --dtype mixed
done
done
