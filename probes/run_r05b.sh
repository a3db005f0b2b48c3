#!/bin/bash
# round 4, call B: X kernel parity + per-shape A/B
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05b
timeout 900 python -m pytest tests/test_kernels_gpu.py -k "x_kernel" -x -q -m gpu > gpurun_out/r05b/pytest_x.log 2>&1; echo "rc=$?" >> gpurun_out/r05b/pytest_x.log
timeout 900 python probes/bench_ntx.py all > gpurun_out/r05b/bench_ntx.log 2>&1
tail -n 6 gpurun_out/r05b/pytest_x.log | cut -c1-300; cat gpurun_out/r05b/bench_ntx.log
