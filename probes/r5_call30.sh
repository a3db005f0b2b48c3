#!/bin/bash
set -x
OUT=gpurun_out/r5c30
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for v in auto 4,8 none; do
  if [ $v = auto ]; then unset X2_SEG_VISION_CUT; unset X2_SEG_VISION_WGRAD; elif [ $v = none ]; then export X2_SEG_VISION_WGRAD=0; unset X2_SEG_VISION_CUT; else export X2_SEG_VISION_CUT=$v; fi
  for c in region mixed; do
    timeout 200 python bench.py --config $c --no-cpu-baseline --no-other-configs --steps 15 --warmup 4 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$c cuts=$v', d['ms_per_step'], d['ms_per_step_spread']['min'])" | tee -a $OUT/ab_cuts_region.txt
  done
done
unset X2_SEG_VISION_CUT; unset X2_SEG_VISION_WGRAD
timeout 600 python -m pytest tests/test_graph_gpu.py tests/test_ddp_gpu.py -q -m gpu -x > $OUT/pytest_graph.log 2>&1; echo "rc=$?" >> $OUT/pytest_graph.log; tail -3 $OUT/pytest_graph.log | cut -c1-200
python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver_cmd.json 2> $OUT/bench_driver_cmd.err
cut -c1-400 $OUT/bench_driver_cmd.json
X2_SEG_TIMES=1 timeout 300 python bench.py --steps 20 --warmup 5 --no-other-configs --no-cpu-baseline > $OUT/bench_segtimes.json 2> $OUT/bench_segtimes.err
grep "segment times" $OUT/bench_segtimes.err | cut -c1-1500
