#!/bin/bash
set -e
mkdir -p gpurun_out/r02
for spec in "1024 0" "512 0" "512 12" "512 24" "512 36" "256 0"; do
  set -- $spec
  MGX_TILE_MAX_N=$1 MGX_FUSE_ROWS=$2 python bench.py --no-cpu-baseline --level 10 --steps 20 --warmup 3 > gpurun_out/r02/bench_x_$1_$2.json 2>/dev/null
  python - "gpurun_out/r02/bench_x_$1_$2.json" "tile<=$1 rows $2" <<'PY'
import json,sys
d=json.load(open(sys.argv[1]))
print(sys.argv[2], round(d["ms_per_step"],4), {k:round(v,4) for k,v in d["phase_ms_per_step"].items() if k in ("smooth_fine","coarse_levels")}, round(d["roofline"]["avg_launch_ms"],4))
PY
done
