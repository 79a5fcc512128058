#!/bin/bash
# needs the debug build: make -C multigrid_nikhil_c-_amd/csrc trace
export MGX_LIBMGX_PATH=$PWD/multigrid_nikhil_c-_amd/libmgx_trace.so
for L in ${*:-12 11 13}; do python3 tools/wave_trace.py $L; done
