#!/bin/bash
timeout -k 10 600 python -m pytest tests/test_gpu_dist.py -x -q -m gpu -k "config4" 2>&1 | tail -3
for r in 1 2; do
echo -n "pair off: "; MGX_PAIR=0 python tools/slab_budget.py 2>&1 | tail -2 | tr '\n' ' '; echo
echo -n "pair on : "; python tools/slab_budget.py 2>&1 | tail -2 | tr '\n' ' '; echo
done
