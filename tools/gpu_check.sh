#!/bin/bash
# a subset of the GPU suite:  gpurun -- bash tools/gpu_check.sh "<pytest args>"
mkdir -p gpurun_out/check
timeout -k 10 1100 python -m pytest $1 -x -q -m gpu > gpurun_out/check/pytest.log 2>&1 || { tail -50 gpurun_out/check/pytest.log; exit 1; }
tail -3 gpurun_out/check/pytest.log
