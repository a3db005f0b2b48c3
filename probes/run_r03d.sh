#!/bin/bash
# round 3, call D: full GPU suite with the parity dump (measured deviations for the tolerance table), serialized kernel trace
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03d
X2_PARITY_DUMP=$GRAFT_REPO_ROOT/gpurun_out/r03d/parity timeout 1800 python -m pytest tests -q -m gpu --maxfail=8 > gpurun_out/r03d/pytest_all.log 2>&1; echo "rc=$?" >> gpurun_out/r03d/pytest_all.log
tail -n 12 gpurun_out/r03d/pytest_all.log | cut -c1-300
bash probes/run_prof.sh r03d base > gpurun_out/r03d/prof.log 2>&1
head -45 gpurun_out/r03d/base_serialized_kernel_stats.txt
