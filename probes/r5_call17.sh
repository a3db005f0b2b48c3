#!/bin/bash
set -x
OUT=gpurun_out/r5c17
mkdir -p $OUT
timeout 1200 python probes/ab_step.py --config base --variants "cut4_8:" "cut5_8:X2_SEG_VISION_CUT=5+8" "cut5_9:X2_SEG_VISION_CUT=5+9" "cut6_9:X2_SEG_VISION_CUT=6+9" "cut6:X2_SEG_VISION_CUT=6" "cut4_7:X2_SEG_VISION_CUT=4+7" --rounds 3 --steps 20 > $OUT/ab_cuts_base.txt 2>&1
grep -v amdgpu $OUT/ab_cuts_base.txt | tail -7
