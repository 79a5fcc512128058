#!/bin/bash
timeout -k 10 900 python -m pytest tests/test_gpu_operators.py tests/test_gpu_fma.py tests/test_gpu_slabs.py tests/test_gpu_solve.py -x -q -m gpu 2>&1 | tail -3
STEPS=50 bash tools/gpu_abc.sh "tools/ab/libmgx_d32.so" 13 12
STEPS=50 bash tools/gpu_abc.sh "tools/ab/libmgx_d32.so" 13 14
