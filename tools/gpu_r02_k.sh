set -e
mkdir -p gpurun_out/r02
python -m pytest tests/test_gpu_dist.py -x -q > gpurun_out/r02/pytest_k.log 2>&1 || { tail -60 gpurun_out/r02/pytest_k.log; exit 1; }
tail -3 gpurun_out/r02/pytest_k.log
MGX_DIST_SINGLE_DEVICE=1 MGX_DIST_BACKEND=gloo python bench.py --gpus 2 --level 13 --steps 3 --warmup 1 > gpurun_out/r02/rehearse2b.json 2> gpurun_out/r02/rehearse2b.err || { tail -30 gpurun_out/r02/rehearse2b.err; exit 1; }
python -c "
import json; d=json.load(open('gpurun_out/r02/rehearse2b.json')); print('rehearsal n_gpus', d['n_gpus'], d['ms_per_step'], d['halo_exchanges_per_step'], d['vcycles_to_1e-8'], d['config']['cut_level'])"
