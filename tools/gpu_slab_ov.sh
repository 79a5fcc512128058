#!/bin/bash
for ov in 1 0 1 0; do echo "MGX_DIST_OVERLAP=$ov"; MGX_DIST_OVERLAP=$ov python3 tools/slab_budget.py 14 2>&1 | grep -E "P=8|P=4"; done
