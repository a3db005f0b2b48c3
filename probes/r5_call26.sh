#!/bin/bash
set -x
OUT=gpurun_out/r5c26
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -q -m gpu > $OUT/pytest_gpu.log 2>&1; echo "rc=$?" >> $OUT/pytest_gpu.log
grep -v "UserWarning\|Consider using\|return Variable\|^$\|amdgpu.ids" $OUT/pytest_gpu.log | tail -n 12 | cut -c1-300
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
