#!/bin/bash
mkdir -p gpurun_out/b10
echo "== alternating issue priority between the two waves of a SIMD (other = tools/ab/libmgx_prio.so) against the default (new)"
bash tools/gpu_ab.sh tools/ab/libmgx_prio.so 13 12 14
echo "== bit check of the variant"
MGX_LIBMGX_PATH=$PWD/tools/ab/libmgx_prio.so timeout -k 10 600 python -m pytest tests/test_gpu_fma.py -x -q -m gpu 2>&1 | tail -2
