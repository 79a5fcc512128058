#!/bin/bash
set -e
mkdir -p gpurun_out/ab
for pct in 23 15 19 27 31 36; do
  for L in 13 12 11; do
    MGX_EDGE_PCT=$pct python bench.py --no-cpu-baseline --level $L --steps 20 --warmup 3 > gpurun_out/ab/bench_pct_${pct}_$L.json 2>/dev/null
    python - "gpurun_out/ab/bench_pct_${pct}_$L.json" "pct$pct L$L" <<'PY'
import json,sys
d=json.load(open(sys.argv[1]))
print(sys.argv[2], round(d["ms_per_step"],4), {k:round(v,4) for k,v in d["phase_ms_per_step"].items() if k in ("smooth_fine","coarse_levels")}, round(d["roofline"]["avg_launch_ms"],4))
PY
  done
done
