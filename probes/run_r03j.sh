#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03j
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -m gpu -k "attention" > gpurun_out/r03j/pytest_attn.log 2>&1; echo "rc=$?" >> gpurun_out/r03j/pytest_attn.log
tail -n 4 gpurun_out/r03j/pytest_attn.log | cut -c1-300
timeout 600 python probes/bench_attn.py > gpurun_out/r03j/bench_attn.log 2>&1; head -14 gpurun_out/r03j/bench_attn.log; tail -n 2 gpurun_out/r03j/bench_attn.log
