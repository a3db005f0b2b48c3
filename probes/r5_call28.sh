#!/bin/bash
set -x
OUT=gpurun_out/r5c28
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 900 python probes/ab_step.py --config base --variants "c4_8:" "c8:X2_SEG_VISION_CUT=8" "c6:X2_SEG_VISION_CUT=6" "c9:X2_SEG_VISION_CUT=9" "c10:X2_SEG_VISION_CUT=10" "c6_9:X2_SEG_VISION_CUT=6+9" "c5_10:X2_SEG_VISION_CUT=5+10" "nodefer:X2_SEG_VISION_WGRAD=0" --rounds 3 --steps 20 > $OUT/ab_cuts_base.txt 2>&1; tail -10 $OUT/ab_cuts_base.txt
