#!/bin/bash
# round 5, GPU call 1: new parity branches + rounded-oracle deviations (measuring run), multi-rank additions, reserved-CU A/B
set -x
OUT=gpurun_out/r5c1
mkdir -p $OUT
export X2_PARITY_DUMP=$OUT/parity X2_PARITY_NO_ASSERT=1
timeout 900 python -m pytest tests/test_model_gpu.py -x -q -s -k "tiny or shallow or base_region" > $OUT/model_parity.log 2>&1
echo "rc model parity $?" >> $OUT/summary.txt
unset X2_PARITY_NO_ASSERT X2_PARITY_DUMP
timeout 600 python -m pytest tests/test_graph_gpu.py -x -q -k "text_only or text_part or mixed" > $OUT/graph_tests.log 2>&1
echo "rc graph $?" >> $OUT/summary.txt
timeout 900 python -m pytest tests/test_ddp_gpu.py -x -q -k "many_ranks" > $OUT/ddp_tests.log 2>&1
echo "rc ddp $?" >> $OUT/summary.txt
timeout 600 python -m pytest tests/test_kernels_gpu.py -x -q -k "gemm" > $OUT/kernel_tests.log 2>&1
echo "rc kernels $?" >> $OUT/summary.txt
for r in 0 16 32 0; do
  X2_RESERVED_CUS=$r timeout 300 python bench.py --steps 20 --warmup 5 --no-other-configs --no-cpu-baseline > $OUT/bench_reserved_$r.$RANDOM.json 2>$OUT/bench_err_$r.log
done
tail -3 $OUT/*.log
grep -h ms_per_step $OUT/bench_reserved_* | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print(d['env_switches'].get('X2_RESERVED_CUS'), d['ms_per_step'], d['ms_per_step_spread'], d['x2_tune_non_default'])
"
