#!/bin/bash
mkdir -p gpurun_out/b13
timeout -k 10 900 python -m pytest tests/test_gpu_dist.py tests/test_gpu_slabs.py tests/test_gpu_fma.py tests/test_gpu_var.py tests/test_gpu_solve.py -x -q -m gpu > gpurun_out/b13/pytest.log 2>&1 || { tail -40 gpurun_out/b13/pytest.log; exit 1; }
tail -2 gpurun_out/b13/pytest.log
echo "== 8-slab budget: tile kernel on small slab ranges (MGX_SLAB_TILE_POINTS; 0 = marching only)"
for t in 1048576 0 2500000 1048576; do echo "tile points $t"; MGX_SLAB_TILE_POINTS=$t python tools/slab_budget.py 14 fma 2>&1 | grep "P=4\|P=8"; done
echo "== exchange / compute overlap with the bands through the tile kernel"
for o in 0 1; do echo "overlap $o"; MGX_DIST_OVERLAP=$o python tools/slab_budget.py 14 fma 2>&1 | grep "P=8"; done
MGX_DIST_OVERLAP=1 MGX_SLAB_TILE_POINTS=0 python tools/slab_budget.py 14 fma 2>&1 | grep "P=8"
