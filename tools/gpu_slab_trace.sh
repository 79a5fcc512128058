#!/bin/bash
# kernel trace of 8 slabs of 16384^2 on one device with kernels serialised (AMD_SERIALIZE_KERNEL=3), so that each
# launch's duration is its own: the per-GPU budget of DESIGN.md 7 by level
set -e
O=gpurun_out/slab
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export AMD_SERIALIZE_KERNEL=3
rocprofv3 --kernel-trace --output-format csv -d $O/kt -o k -- python3 tools/slab_trace.py 8 > /dev/null 2> $O/err.txt
python3 tools/prof_summary.py kt $(find $O/kt -name 'k_kernel_trace.csv') | head -45
