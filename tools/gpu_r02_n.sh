#!/bin/bash
# chunk-height sweep at 8192^2 around whole numbers of 512-workgroup rounds
set -e
mkdir -p gpurun_out/r02
for r in 0 76 80 84 88 100 104 108 112 160 164 168 172 176 184; do
  MGX_FUSE_ROWS=$r python bench.py --no-cpu-baseline --steps 20 --warmup 3 > gpurun_out/r02/bench_n_$r.json 2>/dev/null
  python - "gpurun_out/r02/bench_n_$r.json" "r$r" <<'PY'
import json,sys
d=json.load(open(sys.argv[1]))
print(sys.argv[2], round(d["ms_per_step"],4), {k:round(v,4) for k,v in d["phase_ms_per_step"].items() if k in ("smooth_fine","coarse_levels")}, round(d["roofline"]["avg_launch_ms"],4))
PY
done
