#!/bin/bash
# The round's evidence, one gpurun call:  gpurun --timeout 1200 -- bash tools/gpu_profile.sh <tag>
#   1. rocprofv3 --kernel-trace --stats of the default bench.py command
#   2. PMC passes of the same command, each in its own run (FETCH_SIZE; WRITE_SIZE; two SQ/TCC sets)
#   3. the default bench.py run (with the CPU baseline) and the BASELINE-config table
# Outputs under gpurun_out/<tag>/, condensed there with tools/prof_summary.py / tools/sq_summary.py; copy what is to be
# judged into profiles/.  (The GPU test suite is tools/gpu_suite.sh.)
TAG=${1:-prof}
O=gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o k -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/kt.json 2> $O/kt.err || { tail $O/kt.err; exit 1; }
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/fetch -o f -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2>&1 || exit 1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/write -o f -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2>&1 || exit 1
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $O/sq1 -o s -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2>&1 || exit 1
rocprofv3 --pmc SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/sq2 -o s -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2>&1 || exit 1
echo "profiles done"
python3 tools/prof_summary.py kt $(find $O/kt -name 'k_kernel_trace.csv') > $O/kernel_trace_summary.md
python3 tools/prof_summary.py pmc $(find $O/fetch -name '*counter_collection.csv') $(find $O/write -name '*counter_collection.csv') > $O/pmc_summary.md
python3 tools/sq_summary.py $(find $O/sq1 -name '*kernel_trace.csv') $(find $O/sq1 -name '*counter_collection.csv') $(find $O/sq2 -name '*counter_collection.csv') > $O/sq_summary.md 2> $O/sq_summary.err
head -12 $O/kernel_trace_summary.md
python3 bench.py > $O/bench_n1.json 2> $O/bench_n1.err || { tail $O/bench_n1.err; exit 1; }
python3 -c "
import json;d=json.load(open('$O/bench_n1.json'));print('bench', d['ms_per_step'], d['value'], d['roofline']['frac'], d['cpu_baseline']['value'], d['runtime_libs'])"
python3 tools/run_configs.py > $O/configs.md 2> $O/configs.err
tail -4 $O/configs.md
