#!/bin/bash
# the BASELINE-config table:  gpurun -- bash tools/gpu_configs.sh <tag>   -> gpurun_out/<tag>/configs.md
O=gpurun_out/${1:-cfg}; mkdir -p $O
python3 tools/run_configs.py > $O/configs.md 2> $O/configs.err || { tail $O/configs.err; exit 1; }
cat $O/configs.md
