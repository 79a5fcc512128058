#!/bin/bash
# parity suites first, then tools/gpu_ab.sh (new library against tools/ab/libmgx_prev.so)
mkdir -p gpurun_out/ab
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > gpurun_out/ab/tests.log 2>&1 || { tail -40 gpurun_out/ab/tests.log; exit 1; }
tail -2 gpurun_out/ab/tests.log
bash tools/gpu_ab.sh "$@"
