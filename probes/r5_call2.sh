#!/bin/bash
# round 5, GPU call 2: rounded-oracle deviations with the kernel-faithful attention roundings, text / mixed graph tests,
# LayerNorm forward rows-per-wave A/B (isolated + in the replayed step), NT 256-column kernel phase times
set -x
OUT=gpurun_out/r5c2
mkdir -p $OUT
export X2_PARITY_DUMP=$OUT/parity X2_PARITY_NO_ASSERT=1
timeout 900 python -m pytest tests/test_model_gpu.py -x -q -s -k "tiny or shallow or base_region" > $OUT/model_parity.log 2>&1
echo "rc model parity $?" >> $OUT/summary.txt
unset X2_PARITY_NO_ASSERT X2_PARITY_DUMP
timeout 600 python -m pytest tests/test_graph_gpu.py -x -q -k "text_only or text_part or mixed" > $OUT/graph_tests.log 2>&1
echo "rc graph $?" >> $OUT/summary.txt
timeout 300 python -m pytest tests/test_kernels_gpu.py -x -q -k "layernorm" > $OUT/ln_tests.log 2>&1
echo "rc ln tests $?" >> $OUT/summary.txt
timeout 300 python probes/bench_ln_fwd.py > $OUT/ln_fwd_rows_per_wave.txt 2>&1
X2VLM_HIP_LIB=probes/_probe/libx2vlm_hip_probe.so timeout 300 python probes/nt_phase_times.py > $OUT/nt256_phase_times.txt 2>&1
timeout 900 python probes/ab_step.py --config base --variants "ln_auto:" "ln_rpw1:13=1" "ln_rpw2:13=2" "ln_rpw4:13=4" --rounds 3 --steps 20 > $OUT/ab_ln_step.txt 2>&1
cat $OUT/summary.txt $OUT/ln_fwd_rows_per_wave.txt $OUT/nt256_phase_times.txt $OUT/ab_ln_step.txt | grep -v amdgpu.ids | tail -60
