#!/bin/bash
# round 4, call D: mixed replay, single-rank RCCL through segments, checkpoint on the device
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05d
timeout 900 python -m pytest tests/test_graph_gpu.py tests/test_checkpoint_gpu.py -x -q -m gpu > gpurun_out/r05d/pytest_graph.log 2>&1; echo "rc=$?" >> gpurun_out/r05d/pytest_graph.log
timeout 1500 python -m pytest tests/test_ddp_gpu.py -x -q -m gpu -k "rccl or mixed or plans" > gpurun_out/r05d/pytest_ddp.log 2>&1; echo "rc=$?" >> gpurun_out/r05d/pytest_ddp.log
tail -n 15 gpurun_out/r05d/pytest_graph.log | cut -c1-600; tail -n 15 gpurun_out/r05d/pytest_ddp.log | cut -c1-600
