#!/bin/bash
set -x
OUT=gpurun_out/r5c15
mkdir -p $OUT
timeout 600 python probes/bench_nt_choice.py all > $OUT/nt_choice.txt 2>&1
grep -v amdgpu $OUT/nt_choice.txt
timeout 300 python -m pytest tests/test_kernels_gpu.py -x -q -k "ring_variants" > $OUT/ring_tests.log 2>&1; tail -2 $OUT/ring_tests.log
