#!/bin/bash
mkdir -p gpurun_out/b6
timeout -k 10 600 python -m pytest tests/test_gpu_operators.py tests/test_gpu_solve.py -x -q -m gpu > gpurun_out/b6/pytest.log 2>&1 || { tail -40 gpurun_out/b6/pytest.log; exit 1; }
tail -2 gpurun_out/b6/pytest.log
echo "== four rows in flight in the deep interior bodies (tools/ab/libmgx_pfd4.so = other) against two / three (new)"
bash tools/gpu_ab.sh tools/ab/libmgx_pfd4.so 13 12 14
echo "== mixed"
BENCH_FLAGS="--dtype mixed" bash tools/gpu_ab.sh tools/ab/libmgx_pfd4.so 13
