#!/bin/bash
# the whole GPU suite + smoke + (optionally) the rehearsal:  bash tools/gpu_suite.sh [rehearse]
mkdir -p gpurun_out/suite
timeout -k 10 1100 python -m pytest tests -x -q -m gpu > gpurun_out/suite/pytest_gpu.log 2>&1 || { tail -60 gpurun_out/suite/pytest_gpu.log; exit 1; }
tail -3 gpurun_out/suite/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 || exit 1
if [ "$1" = rehearse ]; then bash tools/gpu_rehearse.sh 4 || exit 1; fi
