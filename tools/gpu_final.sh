#!/bin/bash
# what the driver runs at round end: the GPU suite, smoke(), the default bench; plus the 4-rank staged rehearsal
mkdir -p gpurun_out/final
timeout -k 10 1100 python -m pytest tests/ -x -q -m gpu > gpurun_out/final/pytest_gpu.log 2>&1 || { tail -40 gpurun_out/final/pytest_gpu.log; exit 1; }
tail -2 gpurun_out/final/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 || exit 1
python bench.py > gpurun_out/final/bench.json 2> gpurun_out/final/bench.err || { tail gpurun_out/final/bench.err; exit 1; }
python -c "
import json;d=json.load(open('gpurun_out/final/bench.json'));print('bench', d['n_gpus'], d['ms_per_step'], d['value'], d['roofline']['frac'], d['roofline']['avg_launch_ms'], d['roofline']['traffic'], d['cpu_baseline']['value'], d['ms_per_step_first_5_cycles_from_a_random_guess'])"
bash tools/gpu_rehearse.sh 4
