#!/bin/bash
set -x
OUT=gpurun_out/r5c11
mkdir -p $OUT
X2VLM_HIP_LIB=$PWD/probes/_probe/libx2vlm_hip_ft3.so timeout 300 python -m pytest tests/test_kernels_gpu.py -x -q -k "attention" > $OUT/attn_tests_ft3.log 2>&1
echo "rc attention tests ft3 $?" >> $OUT/summary.txt
for i in 1 2 3; do for v in base ft3 ft2; do
  L=""; [ $v != base ] && L=$PWD/probes/_probe/libx2vlm_hip_$v.so
  X2VLM_HIP_LIB=$L timeout 300 python bench.py --steps 20 --warmup 5 --no-other-configs --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$v', d['ms_per_step'], d['ms_per_step_spread']['median'])" >> $OUT/step_ab.txt
done; done
cat $OUT/summary.txt; cat $OUT/step_ab.txt
