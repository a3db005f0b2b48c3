#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03o
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -m gpu -k "attention or relpos" > gpurun_out/r03o/pytest_attn.log 2>&1; echo "rc=$?" >> gpurun_out/r03o/pytest_attn.log
tail -n 4 gpurun_out/r03o/pytest_attn.log | cut -c1-300
timeout 600 python probes/bench_attn.py > gpurun_out/r03o/bench_attn_log2.log 2>&1; sed -n 5,12p gpurun_out/r03o/bench_attn_log2.log
X2_BENCH_BIAS_LOG2=0 timeout 600 python probes/bench_attn.py > gpurun_out/r03o/bench_attn_raw.log 2>&1; sed -n 5,12p gpurun_out/r03o/bench_attn_raw.log
timeout 600 python -m pytest tests/test_model_gpu.py -q -m gpu -k "tiny or base_shallow or large_shallow or base_region" > gpurun_out/r03o/pytest_model.log 2>&1; echo "rc=$?" >> gpurun_out/r03o/pytest_model.log; tail -n 3 gpurun_out/r03o/pytest_model.log | cut -c1-200
