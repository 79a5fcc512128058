#!/bin/bash
# A/B of the in-tree libmgx.so against another build of it inside ONE GPU call (box-to-box variation is ~5 %):
#   bash tools/gpu_ab.sh <other.so> [level ...]        (default levels 13 12; extra bench.py flags in $BENCH_FLAGS)
# alternating new / other / new / other; prints ms per step, the finest-level and coarse-level parts, ms per pass
mkdir -p gpurun_out/ab
OTHER=$1; shift
LEVELS=${*:-13 12}
for v in new other new other; do
  if [ $v = other ]; then export MGX_LIBMGX_PATH=$PWD/$OTHER; else unset MGX_LIBMGX_PATH; fi
  for L in $LEVELS; do
    python bench.py --no-cpu-baseline --level $L --steps 20 --warmup 3 $BENCH_FLAGS > gpurun_out/ab/bench_${v}_$L.json 2>/dev/null || exit 1
    python - "gpurun_out/ab/bench_${v}_$L.json" "$v L$L" <<'PY'
import json,sys
d=json.load(open(sys.argv[1]))
print(sys.argv[2], round(d["ms_per_step"],4), {k:round(v,4) for k,v in d["phase_ms_per_step"].items() if k in ("smooth_fine","coarse_levels")}, round(d["roofline"]["avg_launch_ms"],4), round(d["roofline"]["frac"],3))
PY
  done
done
