set -e
mkdir -p gpurun_out/r02p
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r02p/kt -o k -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r02p/kt.json 2> gpurun_out/r02p/kt.err
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/r02p/fetch -o f -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/r02p/write -o f -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d gpurun_out/r02p/sq1 -o s -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
rocprofv3 --pmc SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d gpurun_out/r02p/sq2 -o s -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
ls gpurun_out/r02p/*
