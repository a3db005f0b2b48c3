#!/bin/bash
set -x
OUT=gpurun_out/r5c29
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 600 python probes/ab_step.py --config large --variants "c8_16:" "c18:X2_SEG_VISION_CUT=18" "c16:X2_SEG_VISION_CUT=16" "c20:X2_SEG_VISION_CUT=20" "c12_18:X2_SEG_VISION_CUT=12+18" --rounds 2 --steps 8 > $OUT/ab_cuts_large.txt 2>&1; tail -7 $OUT/ab_cuts_large.txt
timeout 600 python probes/ab_step.py --config video --variants "c4_8:" "c9:X2_SEG_VISION_CUT=9" "c8:X2_SEG_VISION_CUT=8" --rounds 3 --steps 20 > $OUT/ab_cuts_video.txt 2>&1; tail -4 $OUT/ab_cuts_video.txt
