#!/bin/bash
# LayerNorm backward at D = 1024 (NV = 4): 3 waves per SIMD (launch bound (256, 3): 154-158 VGPRs, no spills) vs hipcc's free choice (184 VGPRs, 2 waves)
set -x
OUT=gpurun_out/r5c7
mkdir -p $OUT
for i in 1 2; do
  timeout 400 python bench.py --config large --steps 10 --warmup 3 --no-cpu-baseline > $OUT/large_base_$i.json 2>/dev/null
  X2VLM_HIP_LIB=$PWD/probes/_probe/libx2vlm_hip_lnb3.so timeout 400 python bench.py --config large --steps 10 --warmup 3 --no-cpu-baseline > $OUT/large_lnb3_$i.json 2>/dev/null
done
for f in $OUT/large_*.json; do python -c "
import json,sys
d=json.loads([l for l in open('$f') if l.startswith('{')][-1]); print('$f', d['ms_per_step'], d['ms_per_step_spread'])"; done
