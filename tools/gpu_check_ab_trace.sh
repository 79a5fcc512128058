#!/bin/bash
bash tools/gpu_check_ab.sh 13 12 11 || exit 1
bash tools/gpu_wave_trace.sh 2>&1 | grep -E "^L1|duration|interior|edge strips|last waves" | cut -c1-400
