#!/bin/bash
# Rehearsal of `bench.py --gpus N` on the 1-GPU box: N rank processes share device 0 and the halos are
# host-staged through the job's TCP store (explicit opt-in; RCCL refuses two ranks on one device).  Exercises the
# launcher, the rendezvous, mgx_create_rank, the C++ executor and the plans; the timing means nothing.
# MGX_LOG_RUNTIME_LIBS=1: every rank prints the ROCm libraries it has mapped when its handle is created.
mkdir -p gpurun_out/rehearse
N=${1:-4}
export MGX_DIST_SINGLE_DEVICE=1 MGX_DIST_BACKEND=staged MGX_LOG_RUNTIME_LIBS=1
timeout -k 10 500 python bench.py --gpus $N --level 12 --steps 3 --warmup 1 > gpurun_out/rehearse/staged$N.json 2> gpurun_out/rehearse/staged$N.err || { tail -30 gpurun_out/rehearse/staged$N.err; exit 1; }
python - <<PY
import json
d=json.load(open("gpurun_out/rehearse/staged$N.json"))
print("rehearsal", d["n_gpus"], d["ranks_seen_by_collective"], d["transport"], d["ms_per_step"], d["vcycles_to_1e-8"], d["runtime_libs"])
PY
grep -c "runtime libraries mapped" gpurun_out/rehearse/staged$N.err
