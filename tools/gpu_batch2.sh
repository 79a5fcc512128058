#!/bin/bash
# one GPU call: RCCL-after-other-tests diagnosis, the rest of the suite, smoke, rehearsal, fma / separate A/B
mkdir -p gpurun_out/b2
# (1) which HIP call fails inside ncclCommInitRank after test_gpu_api has run in the same process
NCCL_DEBUG=INFO timeout -k 5 300 python -m pytest tests/test_gpu_api.py tests/test_gpu_dist.py -x -q -m gpu -k "test_gpu_api or world_one" > gpurun_out/b2/rccl_after_api.log 2>&1
tail -2 gpurun_out/b2/rccl_after_api.log
grep -n "WARN\|failed\|hipError" gpurun_out/b2/rccl_after_api.log | grep -v "iommu\|Could not read node" | head -12
# (2) the rest of the suite without that test
timeout -k 10 1000 python -m pytest tests -q -m gpu --deselect tests/test_gpu_dist.py::test_rank_handle_with_the_builtin_rccl_transport_at_world_one > gpurun_out/b2/pytest_gpu.log 2>&1
tail -15 gpurun_out/b2/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
bash tools/gpu_rehearse.sh 4
for v in fma separate fma separate; do
  for L in 13 12 14; do
    python bench.py --no-cpu-baseline --arith $v --level $L --steps 20 --warmup 3 > gpurun_out/b2/bench_${v}_$L.json 2>/dev/null || exit 1
    python - "gpurun_out/b2/bench_${v}_$L.json" "$v L$L" <<'PY'
import json,sys
d=json.load(open(sys.argv[1]))
print(sys.argv[2], round(d["ms_per_step"],4), {k:round(v,4) for k,v in d["phase_ms_per_step"].items() if k in ("smooth_fine","coarse_levels")}, round(d["roofline"]["avg_launch_ms"],4), round(d["roofline"]["frac"],3))
PY
  done
done
