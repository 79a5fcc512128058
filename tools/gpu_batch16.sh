#!/bin/bash
STEPS=100 bash tools/gpu_abc.sh "tools/ab/libmgx_d32.so" 13
STEPS=100 bash tools/gpu_abc.sh "tools/ab/libmgx_d32.so" 13
