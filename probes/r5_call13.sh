#!/bin/bash
set -x
OUT=gpurun_out/r5c13
mkdir -p $OUT
timeout 300 python -m pytest tests/test_kernels_gpu.py -x -q -k "gemm" > $OUT/gemm_tests.log 2>&1; echo "rc gemm tests $?" >> $OUT/summary.txt
timeout 300 python probes/bench_nt256_s3.py > $OUT/nt256_s3_per_shape.txt 2>&1
timeout 900 python probes/ab_step.py --config base --variants "ring3:" "ring2:10=1" --rounds 3 --steps 20 > $OUT/ab_s3_base.txt 2>&1
timeout 900 python probes/ab_step.py --config large --variants "ring3:" "ring2:10=1" --rounds 2 --steps 10 > $OUT/ab_s3_large.txt 2>&1
cat $OUT/summary.txt; tail -2 $OUT/gemm_tests.log; grep -v amdgpu $OUT/nt256_s3_per_shape.txt; grep -v amdgpu $OUT/ab_s3_base.txt | tail -3; grep -v amdgpu $OUT/ab_s3_large.txt | tail -3
