#!/bin/bash
# round 5, GPU call 6: NT raster GROUP_M in the replayed step; closing traces (kernel stats base / large / video, PMC bytes)
set -x
OUT=gpurun_out/r5c6
mkdir -p $OUT
timeout 900 python probes/ab_step.py --config base --variants "gm8:" "gm4:0=4" "gm16:0=16" "gm32:0=32" --rounds 3 --steps 20 > $OUT/ab_group_m.txt 2>&1
grep -v amdgpu $OUT/ab_group_m.txt | tail -6
bash probes/run_prof.sh r09a base large video > $OUT/run_prof.log 2>&1
bash probes/run_pmc.sh r09a > $OUT/run_pmc.log 2>&1
tail -12 $OUT/run_pmc.log
