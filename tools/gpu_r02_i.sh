set -e
mkdir -p gpurun_out/r02
run() { tag=$1; shift; env "$@" python bench.py --no-cpu-baseline --steps 6 --warmup 2 > gpurun_out/r02/bench_i_$tag.json 2>/dev/null; python - "gpurun_out/r02/bench_i_$tag.json" "$tag" <<'PY'
import json,sys
d=json.load(open(sys.argv[1]))
print(sys.argv[2], round(d["ms_per_step"],3), {k:round(v,3) for k,v in d["phase_ms_per_step"].items() if k in ("smooth_fine","coarse_levels")}, round(d["roofline"]["avg_launch_ms"],4))
PY
}
for r in 0 96 120 144 156 168 192 240 316; do run r$r MGX_FUSE_ROWS=$r; done
