#!/bin/bash
# why does ncclCommInitRank fail / which librccl serves the process: world-1 communicator on both stacks
mkdir -p gpurun_out/rccl
cat > /tmp/rccl1.py <<'PY'
import os, sys
sys.path.insert(0, os.getcwd())
if os.environ.get("PRELOAD_TORCH"):
    import torch
import __graft_entry__ as ge
pkg = ge.load_package()
print("libs:", pkg.runtime_libs(), flush=True)
try:
    with pkg.Multigrid.rank(0, 1, rccl_id=pkg.rccl_unique_id(), cut_level=8, finest_level=10, coarsest_level=6, mu1=2, mu2=1, schedule=0) as mg:
        mg.fill_rhs(1, 0.0)
        st, h = mg.solve(tol=0.0, max_cycles=2)
        print("OK", h, flush=True)
except Exception as e:
    print("FAILED", e, flush=True)
PY
echo "=== /opt/rocm stack"; NCCL_DEBUG=INFO timeout -k 5 120 python /tmp/rccl1.py > gpurun_out/rccl/sys.log 2>&1; tail -5 gpurun_out/rccl/sys.log
echo "=== torch stack first"; PRELOAD_TORCH=1 NCCL_DEBUG=INFO timeout -k 5 120 python /tmp/rccl1.py > gpurun_out/rccl/torch.log 2>&1; tail -5 gpurun_out/rccl/torch.log
grep -i "warn\|error\|fail" gpurun_out/rccl/sys.log | head -20
ls /opt/rocm/lib/librccl* ; python -c "import torch,os; print(os.listdir(os.path.join(os.path.dirname(torch.__file__),'lib')))" | tr ',' '\n' | grep -i "rccl\|amdhip\|hsa"
