#!/bin/bash
# bench.py --gpus N on a one-GPU box: every rank on device 0.
#  1. host-staged gloo transport (the rehearsal of the multi-process path)
#  2. the RCCL transport, which must fail here (two ranks on one device) and fall back, saying so
mkdir -p gpurun_out/rehearse
MGX_DIST_SINGLE_DEVICE=1 MGX_DIST_BACKEND=gloo timeout -k 10 300 python bench.py --gpus 4 --level 12 --steps 3 --warmup 1 > gpurun_out/rehearse/gloo4.json 2> gpurun_out/rehearse/gloo4.err || { tail -20 gpurun_out/rehearse/gloo4.err; exit 1; }
python - <<'PY'
import json
d=json.load(open('gpurun_out/rehearse/gloo4.json'))
print('gloo4:', d['n_gpus'], round(d['ms_per_step'],3), d['config']['workload'][:200])
PY
MGX_DIST_SINGLE_DEVICE=1 timeout -k 10 300 python bench.py --gpus 2 --level 12 --steps 3 --warmup 1 > gpurun_out/rehearse/rccl2.json 2> gpurun_out/rehearse/rccl2.err; echo "rc=$?"
tail -5 gpurun_out/rehearse/rccl2.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/rehearse/rccl2.json'))
print('rccl2:', d['n_gpus'], round(d['ms_per_step'],3), d['config']['workload'][:260])
PY
