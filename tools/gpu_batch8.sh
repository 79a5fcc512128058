#!/bin/bash
mkdir -p gpurun_out/b8
# correctness of the two-chain (skewed) level schedule first: every parity test that runs the deep passes
timeout -k 10 900 python -m pytest tests/test_gpu_solve.py tests/test_gpu_fma.py tests/test_gpu_dist.py tests/test_gpu_slabs.py -x -q -m gpu > gpurun_out/b8/pytest.log 2>&1 || { tail -40 gpurun_out/b8/pytest.log; exit 1; }
tail -2 gpurun_out/b8/pytest.log
echo "== two level chains per step (new) against the serial chain (other = tools/ab/libmgx_noskew.so)"
bash tools/gpu_ab.sh tools/ab/libmgx_noskew.so 13 12 14
echo "== mixed / f32 / separate"
BENCH_FLAGS="--dtype mixed" bash tools/gpu_ab.sh tools/ab/libmgx_noskew.so 13
BENCH_FLAGS="--arith separate" bash tools/gpu_ab.sh tools/ab/libmgx_noskew.so 13
echo "== general operators"
python tools/var_bench.py 13
