#!/bin/bash
# straight-line (FullTile) bodies in the vision attention backward: ft1 = dQ walk kernel, ft2 = ft1 + the 8-wave LEAN dK/dV kernel
set -x
OUT=gpurun_out/r5c10
mkdir -p $OUT
for v in ft1 ft2; do
  X2VLM_HIP_LIB=$PWD/probes/_probe/libx2vlm_hip_$v.so timeout 300 python -m pytest tests/test_kernels_gpu.py -x -q -k "attention" > $OUT/attn_tests_$v.log 2>&1
  echo "rc attention tests $v $?" >> $OUT/summary.txt
done
for v in base ft1 ft2; do
  L=""; [ $v != base ] && L=$PWD/probes/_probe/libx2vlm_hip_$v.so
  X2VLM_HIP_LIB=$L timeout 300 python probes/bench_attn.py 2>&1 | grep -v amdgpu | grep -E "^vision |vision base" > $OUT/bench_attn_$v.txt
done
for i in 1 2; do for v in base ft1 ft2; do
  L=""; [ $v != base ] && L=$PWD/probes/_probe/libx2vlm_hip_$v.so
  X2VLM_HIP_LIB=$L timeout 300 python bench.py --steps 20 --warmup 5 --no-other-configs --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$v', d['ms_per_step'], d['ms_per_step_spread']['median'])" >> $OUT/step_ab.txt
done; done
cat $OUT/summary.txt; tail -n 3 $OUT/bench_attn_*.txt; cat $OUT/step_ab.txt
