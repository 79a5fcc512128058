#!/bin/bash
export MGX_LIBMGX_PATH=$PWD/multigrid_nikhil_c-_amd/libmgx_trace.so
mkdir -p gpurun_out/trace
echo "== default (two rounds)"; timeout -k 10 200 python tools/wave_trace.py 13 > gpurun_out/trace/r2.txt 2>&1; grep -v "last waves\|decile [0-9]" gpurun_out/trace/r2.txt
echo "== one round of 328-row chunks"; MGX_FUSE_ROWS=328 timeout -k 10 200 python tools/wave_trace.py 13 > gpurun_out/trace/r1.txt 2>&1; grep -v "last waves\|decile [0-9]" gpurun_out/trace/r1.txt
