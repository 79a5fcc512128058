#!/bin/bash
set -e
mkdir -p gpurun_out/r02r
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r02r/kt -o k -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r02r/kt.json 2> gpurun_out/r02r/kt.err
python3 - <<'PY'
import csv,glob,collections
f=glob.glob('gpurun_out/r02r/kt/**/k_kernel_stats.csv',recursive=True)[0]
rows=list(csv.DictReader(open(f)))
tot=sum(float(r['TotalDurationNs']) for r in rows)
for r in rows[:22]:
    print(f"{r['Name'][:70]:70s} calls {int(r['Calls']):6d} avg {float(r['AverageNs'])/1e3:8.1f} us  {100*float(r['TotalDurationNs'])/tot:5.1f}%")
PY
python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/r02r/bench.json
python3 -c "
import json;d=json.load(open('gpurun_out/r02r/bench.json'));print(d['ms_per_step'],d['phase_ms_per_step'],d['roofline']['avg_launch_ms'])"
