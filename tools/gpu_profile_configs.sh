#!/bin/bash
# kernel traces of the other BASELINE configurations (bench.py flags as in tools/run_configs.py):
#   gpurun -- bash tools/gpu_profile_configs.sh <tag>     -> gpurun_out/<tag>/{mixed,f32,rbgs,l14}_kernel_trace_summary.md
TAG=${1:-cfg}
O=gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
run() { name=$1; shift
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/$name -o k -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline "$@" > $O/$name.json 2> $O/$name.err || { tail $O/$name.err; exit 1; }
  python3 tools/prof_summary.py kt $(find $O/$name -name 'k_kernel_trace.csv') > $O/${name}_kernel_trace_summary.md
  echo "== $name: $(python3 -c "import json;d=json.load(open('$O/$name.json'));print(round(d['ms_per_step'],4),'ms per step')")"; head -8 $O/${name}_kernel_trace_summary.md; }
run mixed --level 13 --dtype mixed
run f32 --level 13 --dtype f32
run rbgs --level 13 --smoother rbgs --mu1 2 --mu2 1
run l14 --level 14
