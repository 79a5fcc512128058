#!/bin/bash
mkdir -p gpurun_out/b5
timeout -k 10 600 python -m pytest tests/test_gpu_dist.py tests/test_gpu_slabs.py tests/test_gpu_fma.py -x -q -m gpu > gpurun_out/b5/pytest.log 2>&1 || { tail -40 gpurun_out/b5/pytest.log; exit 1; }
tail -2 gpurun_out/b5/pytest.log
echo "== per-GPU budget, implicit zero guesses on the slab levels"
python tools/slab_budget.py 14 fma 2>&1 | grep "P=1\|P=8"
echo "== ... and at least two rounds of workgroups per slab pass"
MGX_MIN_ROUNDS=2 python tools/slab_budget.py 14 fma 2>&1 | grep "P=1\|P=8"
MGX_MIN_ROUNDS=2 MGX_MIN_ROUNDS_ROWS=600 python tools/slab_budget.py 14 fma 2>&1 | grep "P=8"
b() { python bench.py --no-cpu-baseline --steps 20 --warmup 3 "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print(round(d['ms_per_step'],4), {k:round(v,4) for k,v in d['phase_ms_per_step'].items() if k in ('smooth_fine','coarse_levels')}, round(d['roofline']['avg_launch_ms'],4), round(d['roofline']['frac'],3), round(d['ms_per_step_first_5_cycles_from_a_random_guess'],4))"; }
echo "== depth of the folded passes on the levels below 8192^2 (MGX_FOLD_KMAX)"
for k in 10 8 6 5 10; do echo -n "kmax $k: "; MGX_FOLD_KMAX=$k b --level 13; done
echo "== 4096^2 as the finest level"
for k in 10 5; do echo -n "kmax $k: "; MGX_FOLD_KMAX=$k b --level 12; done
