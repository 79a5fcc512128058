#!/bin/bash
# parity of the deep passes with the pre-scaled rhs ring, then A/B against the build without it
timeout -k 10 900 python -m pytest tests/test_gpu_operators.py tests/test_gpu_fma.py tests/test_gpu_slabs.py tests/test_gpu_solve.py -x -q -m gpu 2>&1 | tail -3
bash tools/gpu_ab.sh tools/ab/libmgx_d32.so 13 12 14
