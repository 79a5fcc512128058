#!/bin/bash
set -e
O=gpurun_out/slab
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --output-format csv -d $O/kt -o k -- python3 tools/slab_trace.py 8 > /dev/null 2> $O/err.txt
python3 tools/prof_summary.py kt $(find $O/kt -name 'k_kernel_trace.csv') | head -40
