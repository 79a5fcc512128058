#!/bin/bash
# per-kernel durations of the two finest-level passes at several pair ratios (rocprofv3 kernel trace of bench.py)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for round in 1 2; do
for ratio in 115 130 145 160; do
  O=gpurun_out/x2_$ratio
  rm -rf $O; mkdir -p $O
  export MGX_PAIR_RATIO=$ratio
  rocprofv3 --kernel-trace --output-format csv -d $O/kt -o k -- python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline > /dev/null 2> $O/err.txt || { tail $O/err.txt; exit 1; }
  echo "== ratio $ratio"
  python3 tools/prof_summary.py kt $(find $O/kt -name 'k_kernel_trace.csv') | grep "k_jacobi_cycle<double, 10, [01], [12], 0, 1> | 5[0-9][0-9]" | grep -v "(b)" | cut -d'|' -f2-7
done
done
