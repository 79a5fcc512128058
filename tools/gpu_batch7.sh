#!/bin/bash
mkdir -p gpurun_out/b7
b() { python bench.py --no-cpu-baseline --steps 20 --warmup 3 "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print(round(d['ms_per_step'],4), {k:round(v,4) for k,v in d['phase_ms_per_step'].items() if k in ('smooth_fine','coarse_levels')}, round(d['roofline']['avg_launch_ms'],4), round(d['roofline']['frac'],3))"; }
echo "== largest grid smoothed by the register-tile kernel (MGX_TILE_MAX_N); 8192^2 cycle and the reference's own hierarchy 10..7"
for t in 1024 512 256 0 1024; do echo -n "tile_max_n $t: L13 "; MGX_TILE_MAX_N=$t b --level 13; echo -n "                 L10 "; MGX_TILE_MAX_N=$t b --level 10; done
