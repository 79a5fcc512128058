#!/bin/bash
# round 2: FMG on slabs + whole GPU suite
mkdir -p gpurun_out/r02
timeout -k 10 900 python -m pytest tests/test_gpu_dist.py -x -q -m gpu > gpurun_out/r02/m_dist.log 2>&1 || { tail -30 gpurun_out/r02/m_dist.log; exit 1; }
tail -3 gpurun_out/r02/m_dist.log
timeout -k 10 900 python -m pytest tests -x -q -m gpu --deselect tests/test_gpu_dist.py > gpurun_out/r02/m_all.log 2>&1 || { tail -30 gpurun_out/r02/m_all.log; exit 1; }
tail -3 gpurun_out/r02/m_all.log
