set -e
mkdir -p gpurun_out/r02
export MGX_PLAN_PRE=10 MGX_PLAN_POST=10
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d gpurun_out/r02/sq4 -o s -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r02/sq4.json 2> gpurun_out/r02/sq4.err
rocprofv3 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_ACTIVE_INST_VMEM GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d gpurun_out/r02/sq5 -o s -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r02/sq5.json 2> gpurun_out/r02/sq5.err || echo sq5 failed
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/r02/fetch10 -o f -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2>&1 || echo fetch failed
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/r02/write10 -o f -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2>&1 || echo write failed
ls gpurun_out/r02/sq4 gpurun_out/r02/sq5 gpurun_out/r02/fetch10
