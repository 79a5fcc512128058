#!/bin/bash
mkdir -p gpurun_out/b4
# the GPU suite exactly as the driver runs it, then smoke
timeout -k 10 1000 python -m pytest tests/ -x -q -m gpu > gpurun_out/b4/pytest_gpu.log 2>&1 || { tail -40 gpurun_out/b4/pytest_gpu.log; exit 1; }
tail -2 gpurun_out/b4/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
b() { python bench.py --no-cpu-baseline --steps 20 --warmup 3 "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print(round(d['ms_per_step'],4), {k:round(v,4) for k,v in d['phase_ms_per_step'].items() if k in ('smooth_fine','coarse_levels')}, round(d['roofline']['avg_launch_ms'],4), round(d['roofline']['frac'],3), d['vcycles_to_1e-8'])"; }
echo "== chunk height of the deep passes (MGX_FUSE_ROWS), 8192^2 fma"
for R in 0 108 132 204 252 328; do echo -n "rows $R: "; MGX_FUSE_ROWS=$R b --level 13; done
echo -n "rows 0 again: "; b --level 13
echo "== config 5 (mixed), V(10,10): 10-level float passes (new) vs capped at 8 (old)"
for k in 10 8 10 8; do echo -n "kmax $k: "; MGX_FOLD_KMAX=$k MGX_FOLD_KMAX_BIG=$k MGX_FOLD_KMAX_NOPOST=$k b --level 13 --dtype mixed; done
echo -n "mixed separate: "; b --level 13 --dtype mixed --arith separate
echo -n "f32 fma: "; b --level 13 --dtype f32
echo -n "f32 fma kmax 8: "; MGX_FOLD_KMAX=8 MGX_FOLD_KMAX_BIG=8 MGX_FOLD_KMAX_NOPOST=8 b --level 13 --dtype f32
echo "== per-GPU compute budget at 16384^2"
python tools/slab_budget.py 14 fma
