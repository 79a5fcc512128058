#!/bin/bash
# The round's evidence, one gpurun call:  gpurun --timeout 1200 -- bash tools/gpu_profile.sh <tag>
#   1. the whole GPU test suite
#   2. rocprofv3 --kernel-trace --stats of the default bench.py command
#   3. PMC passes of the same command, each in its own run (FETCH_SIZE; WRITE_SIZE; two SQ/TCC sets)
#   4. the default bench.py run (with the CPU baseline) and the BASELINE-config table
# Outputs under gpurun_out/<tag>/; condense them afterwards with tools/prof_summary.py / tools/sq_summary.py
# and copy what is to be judged into profiles/.
set -e
TAG=${1:-prof}
O=gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 1000 python3 -m pytest tests -x -q -m gpu > $O/pytest_gpu.log 2>&1 || { tail -40 $O/pytest_gpu.log; exit 1; }
tail -2 $O/pytest_gpu.log
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o k -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/kt.json 2> $O/kt.err
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/fetch -o f -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/write -o f -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $O/sq1 -o s -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
rocprofv3 --pmc SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/sq2 -o s -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
echo "profiles done"
python3 bench.py > $O/bench_n1.json 2> $O/bench_n1.err
python3 -c "
import json;d=json.load(open('$O/bench_n1.json'));print('bench', d['ms_per_step'], d['value'], d['roofline']['frac'], d['cpu_baseline']['value'])"
python3 tools/run_configs.py > $O/configs.md 2> $O/configs.err
tail -3 $O/configs.md
