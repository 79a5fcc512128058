#!/bin/bash
# several builds of libmgx.so against the in-tree one inside ONE GPU call:  bash tools/gpu_abc.sh "<lib ...>" [level ...]
# ("new" = the in-tree build; others are paths relative to the repo); two alternating rounds; $BENCH_FLAGS as in gpu_ab.sh
mkdir -p gpurun_out/ab
LIBS=$1; shift
LEVELS=${*:-13 12}
for round in 1 2; do
 for v in new $LIBS; do
  if [ $v = new ]; then unset MGX_LIBMGX_PATH; else export MGX_LIBMGX_PATH=$PWD/$v; fi
  for L in $LEVELS; do
    python bench.py --no-cpu-baseline --level $L --steps ${STEPS:-20} --warmup 3 $BENCH_FLAGS > gpurun_out/ab/bench_abc.json 2>/dev/null || exit 1
    python - gpurun_out/ab/bench_abc.json "$(basename $v) L$L" <<'PY'
import json,sys
d=json.load(open(sys.argv[1]))
print(sys.argv[2], round(d["ms_per_step"],4), {k:round(v,4) for k,v in d["phase_ms_per_step"].items() if k in ("smooth_fine","coarse_levels")}, round(d["roofline"]["avg_launch_ms"],4), round(d["roofline"]["frac"],3))
PY
  done
 done
done
