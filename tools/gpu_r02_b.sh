set -e
mkdir -p gpurun_out/r02
python -m pytest tests/test_gpu_dist.py -x -q > gpurun_out/r02/pytest_dist.log 2>&1 || { tail -60 gpurun_out/r02/pytest_dist.log; exit 1; }
tail -3 gpurun_out/r02/pytest_dist.log
python -m pytest tests -m gpu -x -q --deselect tests/test_gpu_dist.py > gpurun_out/r02/pytest_b.log 2>&1 || { tail -40 gpurun_out/r02/pytest_b.log; exit 1; }
tail -3 gpurun_out/r02/pytest_b.log
MGX_DIST_SINGLE_DEVICE=1 MGX_DIST_BACKEND=gloo python bench.py --gpus 2 --level 13 --steps 3 --warmup 1 > gpurun_out/r02/rehearse2.json 2> gpurun_out/r02/rehearse2.err || { tail -30 gpurun_out/r02/rehearse2.err; exit 1; }
python -c "
import json; d=json.load(open('gpurun_out/r02/rehearse2.json')); print('rehearsal n_gpus', d['n_gpus'], d['ms_per_step'], d['halo_exchanges_per_step'], d['vcycles_to_1e-8'], d['speedup_vs_single_gpu_same_workload'])"
python bench.py --gpus 2 > gpurun_out/r02/gpus2_nodev.json 2> gpurun_out/r02/gpus2_nodev.err || echo "expected failure rc=$? (one device): $(cat gpurun_out/r02/gpus2_nodev.json | cut -c1-300)"
