#!/bin/bash
# round 4, call F: rest of the GPU suite (after test_model_gpu) + NT256 stagger probe
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05f
timeout 600 python probes/bench_nt256_stagger.py > gpurun_out/r05f/stagger.log 2>&1; cat gpurun_out/r05f/stagger.log | grep -v amdgpu.ids
timeout 3000 python -m pytest tests/test_model_gpu.py tests/test_optim_gpu.py tests/test_retrieval.py tests/test_train_mode_gpu.py -x -q -m gpu -k "large_full_b32 or optim or retrieval or train_mode or fifty or adamw or mutating or short" -s > gpurun_out/r05f/pytest_rest.log 2>&1; echo "rc=$?" >> gpurun_out/r05f/pytest_rest.log
grep -v "UserWarning\|Consider using\|return Variable\|^$\|amdgpu.ids" gpurun_out/r05f/pytest_rest.log | tail -n 30 | cut -c1-1200
