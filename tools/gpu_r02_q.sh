#!/bin/bash
# chunk heights that make the finest-level pass ONE round of 512 workgroups
set -e
mkdir -p gpurun_out/r02
for spec in "13 300" "13 324" "13 336" "13 348" "13 372" "12 80" "12 84" "12 90" "12 96" "14 0" "14 200" "14 336" "14 680" "14 1366"; do
  set -- $spec
  MGX_FUSE_ROWS=$2 python bench.py --no-cpu-baseline --level $1 --steps 10 --warmup 3 > gpurun_out/r02/bench_q_$1_$2.json 2>/dev/null
  python - "gpurun_out/r02/bench_q_$1_$2.json" "L$1 r$2" <<'PY'
import json,sys
d=json.load(open(sys.argv[1]))
print(sys.argv[2], round(d["ms_per_step"],4), {k:round(v,4) for k,v in d["phase_ms_per_step"].items() if k in ("smooth_fine","coarse_levels")}, round(d["roofline"]["avg_launch_ms"],4))
PY
done
