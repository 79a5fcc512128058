#!/bin/bash
mkdir -p gpurun_out/rccl
NCCL_DEBUG=INFO timeout -k 5 300 python -m pytest tests/test_gpu_dist.py -x -q -m gpu > gpurun_out/rccl/dist_only.log 2>&1; tail -3 gpurun_out/rccl/dist_only.log
grep -n "WARN\|error" gpurun_out/rccl/dist_only.log | grep -v "iommu\|Could not read node" | head -20
