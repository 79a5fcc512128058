#!/bin/bash
# FMA mode: its tests, then bench.py in both arithmetic modes, alternating, at the levels given
mkdir -p gpurun_out/fma
timeout -k 10 900 python -m pytest tests/test_gpu_fma.py -x -q > gpurun_out/fma/pytest.log 2>&1 || { tail -40 gpurun_out/fma/pytest.log; exit 1; }
tail -2 gpurun_out/fma/pytest.log
LEVELS=${*:-13}
for v in fma separate fma separate; do
  for L in $LEVELS; do
    python bench.py --no-cpu-baseline --arith $v --level $L --steps 20 --warmup 3 > gpurun_out/fma/bench_${v}_$L.json 2>/dev/null || exit 1
    python - "gpurun_out/fma/bench_${v}_$L.json" "$v L$L" <<'PY'
import json,sys
d=json.load(open(sys.argv[1]))
print(sys.argv[2], round(d["ms_per_step"],4), {k:round(v,4) for k,v in d["phase_ms_per_step"].items() if k in ("smooth_fine","coarse_levels")}, round(d["roofline"]["avg_launch_ms"],4), round(d["roofline"]["frac"],3))
PY
  done
done
