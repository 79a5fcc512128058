set -e
mkdir -p gpurun_out/r02
for L in 12 11; do
for r in 0 12 24 36 48 60 72 84 96 108 132 180; do
  MGX_FUSE_ROWS=$r python bench.py --no-cpu-baseline --level $L --steps 20 --warmup 3 > gpurun_out/r02/bench_j_${L}_$r.json 2>/dev/null
  python - "gpurun_out/r02/bench_j_${L}_$r.json" "L$L r$r" <<'PY'
import json,sys
d=json.load(open(sys.argv[1]))
print(sys.argv[2], round(d["ms_per_step"],4), {k:round(v,4) for k,v in d["phase_ms_per_step"].items() if k in ("smooth_fine","coarse_levels")}, round(d["roofline"]["avg_launch_ms"],4), d["roofline"]["launches_timed"])
PY
done; done
