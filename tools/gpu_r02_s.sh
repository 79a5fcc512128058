#!/bin/bash
# pass plans on the 2048^2 and 4096^2 levels (as finest level of a smaller hierarchy)
set -e
mkdir -p gpurun_out/r02
for L in 11 12; do
for plan in "10" "5,5" "6,4" "8,2" "4,3,3" "5,3,2"; do
  tag=$(echo $plan | tr , _)
  MGX_PLAN_MIN_N=2048 MGX_PLAN_PRE=$plan MGX_PLAN_POST=$plan python bench.py --no-cpu-baseline --level $L --steps 20 --warmup 3 > gpurun_out/r02/bench_s_${L}_$tag.json 2>/dev/null
  python - "gpurun_out/r02/bench_s_${L}_$tag.json" "L$L [$plan]" <<'PY'
import json,sys
d=json.load(open(sys.argv[1]))
print(sys.argv[2], round(d["ms_per_step"],4), {k:round(v,4) for k,v in d["phase_ms_per_step"].items() if k in ("smooth_fine","coarse_levels")}, round(d["roofline"]["avg_launch_ms"],4))
PY
done; done
