#!/bin/bash
# SQ counter passes of one bench.py configuration:  gpurun -- bash tools/gpu_sq_config.sh <tag> <bench flags...>
TAG=$1; shift
O=gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $O/sq1 -o s -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline "$@" > /dev/null 2>&1 || exit 1
rocprofv3 --pmc SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/sq2 -o s -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline "$@" > /dev/null 2>&1 || exit 1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/fetch -o f -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline "$@" > /dev/null 2>&1 || exit 1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/write -o f -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline "$@" > /dev/null 2>&1 || exit 1
python3 tools/sq_summary.py $(find $O/sq1 -name '*kernel_trace.csv') $(find $O/sq1 -name '*counter_collection.csv') $(find $O/sq2 -name '*counter_collection.csv') > $O/sq_summary.md
python3 tools/prof_summary.py pmc $(find $O/fetch -name '*counter_collection.csv') $(find $O/write -name '*counter_collection.csv') > $O/pmc_summary.md
head -8 $O/sq_summary.md
grep "HBM bytes" $O/pmc_summary.md | head -5
