set -e
mkdir -p gpurun_out/r02
rocprofv3 -L > gpurun_out/r02/counters_list.txt 2>&1 || true
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d gpurun_out/r02/sq1 -o s -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r02/sq1.json 2> gpurun_out/r02/sq1.err
rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d gpurun_out/r02/sq2 -o s -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r02/sq2.json 2> gpurun_out/r02/sq2.err || echo "sq2 failed"
rocprofv3 --pmc SQ_LEVEL_WAVES SQ_INSTS_VALU SQ_WAVES TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d gpurun_out/r02/sq3 -o s -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r02/sq3.json 2> gpurun_out/r02/sq3.err || echo "sq3 failed"
ls -R gpurun_out/r02 | head -50
