#!/bin/bash
# round 4, call G: staggered 8-phase NT256: parity + per-shape A/B
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05g
X2_TUNE=10=1 timeout 900 python -m pytest tests/test_kernels_gpu.py -k "256_column" -x -q -m gpu > gpurun_out/r05g/pytest.log 2>&1; echo "rc=$?" >> gpurun_out/r05g/pytest.log
tail -n 4 gpurun_out/r05g/pytest.log | cut -c1-300
timeout 900 python probes/bench_nt256_st.py all > gpurun_out/r05g/bench.log 2>&1; grep -v amdgpu.ids gpurun_out/r05g/bench.log
