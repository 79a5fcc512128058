#!/bin/bash
mkdir -p gpurun_out/b9
export MGX_LIBMGX_PATH=$PWD/multigrid_nikhil_c-_amd/libmgx_trace.so
python3 tools/wave_trace.py 13 2>&1 | head -40
unset MGX_LIBMGX_PATH
b() { python bench.py --no-cpu-baseline --steps 20 --warmup 3 "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print(round(d['ms_per_step'],4), {k:round(v,4) for k,v in d['phase_ms_per_step'].items()}, round(d['roofline']['avg_launch_ms'],4))"; }
echo "== rows per wave of the fp64 update + residual pass of the mixed cycle (MGX_MIXED_ROWS)"
for r in 0 1 2 4 0 1; do echo -n "rows $r: "; MGX_MIXED_ROWS=$r b --level 13 --dtype mixed; done
