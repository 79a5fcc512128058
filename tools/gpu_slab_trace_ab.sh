#!/bin/bash
# serialised kernel trace of the 8-slab cycle twice in one call: shortened chunks at the slabs' inner ends (round 3's earlier
# geometry) against interior ends; prints the per-kernel tables and the sum of all kernel time per cycle and GPU
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export AMD_SERIALIZE_KERNEL=3
for v in 0 1; do
  O=gpurun_out/slab_ab$v
  mkdir -p $O
  export MGX_SLAB_ENDS_INTERIOR=$v
  rocprofv3 --kernel-trace --output-format csv -d $O/kt -o k -- python3 tools/slab_trace.py 8 > /dev/null 2> $O/err.txt || { tail $O/err.txt; exit 1; }
  echo "== MGX_SLAB_ENDS_INTERIOR=$v"
  python3 tools/prof_summary.py kt $(find $O/kt -name 'k_kernel_trace.csv') > $O/summary.md
  head -24 $O/summary.md
  python3 - $O/summary.md <<'PY'
import sys
tot=0.0
for l in open(sys.argv[1]):
    c=[x.strip() for x in l.split('|')]
    if len(c)>8 and c[7].replace('.','').isdigit(): tot+=float(c[7])
print("total kernel ms in the trace:", round(tot,3))
PY
done
