#!/bin/bash
# closing set on the final tree (one-pass attention backward, fork-free tail): kernel traces (base / large / video), PMC bytes, the driver's bench
# command, the segment timeline of the replayed base step
set -x
OUT=gpurun_out/r5c27
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
bash probes/run_prof.sh r11a base large video > $OUT/run_prof.log 2>&1
bash probes/run_pmc.sh r11a > $OUT/run_pmc.log 2>&1
tail -12 $OUT/run_pmc.log
python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver_cmd.json 2> $OUT/bench_driver_cmd.err
cut -c1-400 $OUT/bench_driver_cmd.json
X2_SEG_TIMES=1 timeout 300 python bench.py --steps 20 --warmup 5 --no-other-configs --no-cpu-baseline > $OUT/bench_segtimes.json 2> $OUT/bench_segtimes.err
grep "segment times" $OUT/bench_segtimes.err | cut -c1-1500
