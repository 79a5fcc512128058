#!/bin/bash
mkdir -p gpurun_out/b3
# the three tests that failed in batch 2
timeout -k 5 300 python -m pytest tests/test_gpu_slabs.py "tests/test_gpu_dist.py::test_poisson_driver_binary_runs_the_reference_sequence" -x -q -m gpu 2>&1 | tail -3
# RCCL communicator after the other tests of the process: what fails inside ncclCommInitRank
NCCL_DEBUG=WARN timeout -k 5 300 python -m pytest tests/test_gpu_api.py tests/test_gpu_dist.py -x -q -m gpu > gpurun_out/b3/rccl_after_all.log 2>&1
tail -2 gpurun_out/b3/rccl_after_all.log
grep -n "WARN\|hipError\|failed" gpurun_out/b3/rccl_after_all.log | grep -v "iommu\|Could not read node" | head -10
# three waves per SIMD for the deep pre-smoothing kernels (tools/ab/libmgx_w3.so) against the default build
bash tools/gpu_ab.sh tools/ab/libmgx_w3.so 13 12
