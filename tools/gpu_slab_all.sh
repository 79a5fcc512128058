#!/bin/bash
python3 tools/slab_budget.py 14
bash tools/gpu_slab_trace.sh | grep -E "k_jacobi_cycle|workgroups" | cut -c1-110 | head -12
