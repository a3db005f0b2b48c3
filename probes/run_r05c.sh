#!/bin/bash
# round 4, call C: graph.py advice fixes + multi-rank safety tests
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05c
timeout 1500 python -m pytest tests/test_ddp_gpu.py tests/test_graph_gpu.py -x -q -m gpu > gpurun_out/r05c/pytest_ddp.log 2>&1; echo "rc=$?" >> gpurun_out/r05c/pytest_ddp.log
tail -n 25 gpurun_out/r05c/pytest_ddp.log | cut -c1-400
