#!/bin/bash
# the default bench.py line (with the CPU baseline), pretty-printed keys of interest
mkdir -p gpurun_out/bench
python bench.py > gpurun_out/bench/bench.json 2> gpurun_out/bench/bench.err || { tail gpurun_out/bench/bench.err; exit 1; }
python -c "
import json;d=json.load(open('gpurun_out/bench/bench.json'))
print('ms_per_step', d['ms_per_step'], 'value', d['value'], 'frac', d['roofline']['frac'], 'avg_launch_ms', d['roofline']['avg_launch_ms'])
print('profiling_off', d.get('profiling_off'))
print('cpu_baseline', d['cpu_baseline']['value'], d['cpu_baseline']['cores'])
print('first5', d['ms_per_step_first_5_cycles_from_a_random_guess'], 'cycles', d['vcycles_to_1e-8'])"
