#!/bin/bash
# finest-level slab passes (8 slabs of 16384^2, serialised kernel trace) with and without paired chunk heights on slabs
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export AMD_SERIALIZE_KERNEL=3
for v in 200 100 200 100; do
  O=gpurun_out/slab_pair$v
  rm -rf $O; mkdir -p $O
  export MGX_PAIR_MIN_ROWS=$v
  rocprofv3 --kernel-trace --output-format csv -d $O/kt -o k -- python3 tools/slab_trace.py 8 > /dev/null 2> $O/err.txt || { tail $O/err.txt; exit 1; }
  echo "== MGX_PAIR_MIN_ROWS=$v"
  python3 tools/prof_summary.py kt $(find $O/kt -name 'k_kernel_trace.csv') | grep "k_jacobi_cycle" | head -6
done
