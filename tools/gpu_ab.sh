#!/bin/bash
# A/B of multigrid_nikhil_c-_amd/libmgx.so against tools/ab/libmgx_prev.so: bench.py at the levels given
# (default 13 12 11), alternating, twice
mkdir -p gpurun_out/ab
LEVELS=${*:-13 12 11}
for v in new prev new prev; do
  if [ $v = prev ]; then export MGX_LIBMGX_PATH=$PWD/tools/ab/libmgx_prev.so; else unset MGX_LIBMGX_PATH; fi
  for L in $LEVELS; do
    python bench.py --no-cpu-baseline --level $L --steps 20 --warmup 3 > gpurun_out/ab/bench_${v}_$L.json 2>/dev/null || exit 1
    python - "gpurun_out/ab/bench_${v}_$L.json" "$v L$L" <<'PY'
import json,sys
d=json.load(open(sys.argv[1]))
print(sys.argv[2], round(d["ms_per_step"],4), {k:round(v,4) for k,v in d["phase_ms_per_step"].items() if k in ("smooth_fine","coarse_levels")}, round(d["roofline"]["avg_launch_ms"],4))
PY
  done
done
