#!/bin/bash
# the whole GPU suite with extra environment, e.g.  gpurun -- bash tools/gpu_check_env.sh MGX_PAIR_MIN_ROWS=16 MGX_MIN_CHUNK=8
# (every launch of a deep pass then takes the one-round paired geometry wherever it exists: a parity check of csrc/mgx_geom.hpp)
mkdir -p gpurun_out/check
for kv in "$@"; do export "$kv"; done
timeout -k 10 1100 python -m pytest tests -q -m gpu > gpurun_out/check/pytest_env.log 2>&1
rc=$?
tail -15 gpurun_out/check/pytest_env.log
exit $rc
