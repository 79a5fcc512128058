#!/bin/bash
# suite + smoke + rehearsal, then the fma / separate A/B
bash tools/gpu_suite.sh rehearse || exit 1
bash tools/gpu_fma.sh 13 12 14 2>&1 | grep -v "^\.\|passed"
