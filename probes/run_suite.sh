#!/bin/bash
# whole GPU suite on the current tree: usage  gpurun -- 'bash probes/run_suite.sh <tag>'
tag=${1:-suite}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/$tag
timeout 3000 python -m pytest tests -x -q -m gpu > gpurun_out/$tag/pytest_gpu.log 2>&1; echo "rc=$?" >> gpurun_out/$tag/pytest_gpu.log
grep -v "UserWarning\|Consider using\|return Variable\|^$\|amdgpu.ids" gpurun_out/$tag/pytest_gpu.log | tail -n 12 | cut -c1-600
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
