#!/bin/bash
set -x
OUT=gpurun_out/r5c23
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_kernels_gpu.py tests/test_train_mode_gpu.py -q -m gpu -k "attention or bert_layers or whole_model" > $OUT/pytest_attn.log 2>&1; rc=$?; echo "rc=$rc" >> $OUT/pytest_attn.log
grep -v "UserWarning\|Consider using\|return Variable\|^$\|amdgpu.ids" $OUT/pytest_attn.log | tail -n 25 | cut -c1-400
timeout 300 python probes/bench_attn_onepass.py > $OUT/bench_attn_onepass.txt 2>&1; tail -9 $OUT/bench_attn_onepass.txt
if [ $rc -eq 0 ]; then
  timeout 900 python probes/ab_step.py --config base --variants "two:14=1" "one:14=0" "one_noaux:14=0,X2_AUX_OVERLAP=0" "two_noaux:14=1,X2_AUX_OVERLAP=0" --rounds 3 --steps 20 > $OUT/ab_base.txt 2>&1; tail -6 $OUT/ab_base.txt
fi
