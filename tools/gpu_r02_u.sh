#!/bin/bash
# default chunk height against taller ones at 8192^2, alternating (box-to-box clocks differ)
set -e
mkdir -p gpurun_out/r02
for rep in 1 2 3; do
for r in 0 120 168 336; do
  MGX_FUSE_ROWS=$r python bench.py --no-cpu-baseline --steps 20 --warmup 3 > gpurun_out/r02/bench_u_${r}_$rep.json 2>/dev/null
  python - "gpurun_out/r02/bench_u_${r}_$rep.json" "r$r" <<'PY'
import json,sys
d=json.load(open(sys.argv[1]))
print(sys.argv[2], round(d["ms_per_step"],4), {k:round(v,4) for k,v in d["phase_ms_per_step"].items() if k in ("smooth_fine","coarse_levels")}, round(d["roofline"]["avg_launch_ms"],4))
PY
done; done
