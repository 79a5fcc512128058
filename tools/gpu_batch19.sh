#!/bin/bash
export MGX_LIBMGX_PATH=$PWD/multigrid_nikhil_c-_amd/libmgx_trace.so
mkdir -p gpurun_out/trace
echo "== one round of 348-row chunks"; MGX_FUSE_ROWS=344 timeout -k 10 200 python tools/wave_trace.py 13 > gpurun_out/trace/r1b.txt 2>&1; grep -v "last waves\|decile [0-9]\|by cu\|by se\|by simd" gpurun_out/trace/r1b.txt
echo "== default"; timeout -k 10 200 python tools/wave_trace.py 13 > gpurun_out/trace/r2b.txt 2>&1; grep "span\|SIMDs with\|faster\|block index\|hardware wave" gpurun_out/trace/r2b.txt
