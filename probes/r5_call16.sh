#!/bin/bash
set -x
OUT=gpurun_out/r5c16
mkdir -p $OUT
timeout 300 python -m pytest tests/test_kernels_gpu.py -x -q -k "gemm" > $OUT/gemm_tests.log 2>&1; echo "rc gemm tests $?" >> $OUT/summary.txt
timeout 900 python probes/ab_step.py --config base --variants "new_rule:" "old_rule:11=1" --rounds 3 --steps 20 > $OUT/ab_rule_base.txt 2>&1
timeout 900 python probes/ab_step.py --config large --variants "new_rule:" "old_rule:11=1" --rounds 2 --steps 10 > $OUT/ab_rule_large.txt 2>&1
cat $OUT/summary.txt; grep -v amdgpu $OUT/ab_rule_base.txt | tail -3; grep -v amdgpu $OUT/ab_rule_large.txt | tail -3
