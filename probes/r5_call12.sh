#!/bin/bash
set -x
OUT=gpurun_out/r5c12
mkdir -p $OUT
X2VLM_HIP_LIB=$PWD/probes/_probe/libx2vlm_hip_probe.so timeout 300 python probes/nt_loop_ablation.py > $OUT/nt_loop_ablation.txt 2>&1
grep -v amdgpu $OUT/nt_loop_ablation.txt
timeout 300 python -m pytest tests/test_kernels_gpu.py -x -q -k "gemm" > $OUT/gemm_tests.log 2>&1; tail -2 $OUT/gemm_tests.log
