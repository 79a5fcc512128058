#!/bin/bash
# SQ counters of the finest-level passes at two chunk heights (72 = default, 168)
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for r in 0 168; do
  d=gpurun_out/r02t/r$r
  mkdir -p $d
  export MGX_FUSE_ROWS=$r
  rocprofv3 --kernel-trace --output-format csv -d $d/kt -o k -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
  rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $d/sq1 -o s -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
  rocprofv3 --pmc SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $d/sq2 -o s -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
  echo "== MGX_FUSE_ROWS=$r"
  python3 tools/sq_summary.py $(find $d/kt -name 'k_kernel_trace.csv') $(find $d/sq1 -name 's_counter_collection.csv') $(find $d/sq2 -name 's_counter_collection.csv') | grep -E "kernel|k_jacobi_cycle" | head -8
done
