#!/bin/bash
# multi-GPU driver on one device: parity suite, then the per-GPU compute budget (tools/slab_budget.py)
mkdir -p gpurun_out/slab
timeout -k 10 600 python -m pytest tests/test_gpu_dist.py -x -q -m gpu > gpurun_out/slab/tests.log 2>&1 || { tail -30 gpurun_out/slab/tests.log; exit 1; }
tail -2 gpurun_out/slab/tests.log
python3 tools/slab_budget.py 14
