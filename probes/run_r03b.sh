#!/bin/bash
# round 3, call B: NT256 kernel parity + per-shape A/B; segmented graph tests
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03b
timeout 900 python -m pytest tests/test_kernels_gpu.py -k "gemm_nt" -x -q -m gpu > gpurun_out/r03b/pytest_gemm.log 2>&1; echo "rc=$?" >> gpurun_out/r03b/pytest_gemm.log
timeout 600 python -m pytest tests/test_graph_gpu.py -x -q -m gpu > gpurun_out/r03b/pytest_graph.log 2>&1; echo "rc=$?" >> gpurun_out/r03b/pytest_graph.log
timeout 600 python probes/bench_nt256.py all > gpurun_out/r03b/bench_nt256.log 2>&1
tail -n 4 gpurun_out/r03b/pytest_gemm.log gpurun_out/r03b/pytest_graph.log | cut -c1-300; cat gpurun_out/r03b/bench_nt256.log
