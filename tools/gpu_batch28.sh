#!/bin/bash
for r in 1 2 3; do
echo -n "ends shortened: "; MGX_SLAB_ENDS_INTERIOR=0 python tools/slab_budget.py 2>&1 | tail -1
echo -n "ends interior : "; python tools/slab_budget.py 2>&1 | tail -1
done
