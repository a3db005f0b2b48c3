#!/bin/bash
set -x
OUT=gpurun_out/r5c21
mkdir -p $OUT
export TMPDIR=/tmp
rm -rf /tmp/prof_replay
(cd /tmp && rocprofv3 --kernel-trace -d /tmp/prof_replay -o replay -- python $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 2 --no-other-configs --no-cpu-baseline > $GRAFT_REPO_ROOT/$OUT/prof_replay.log 2>&1)
db=$(find /tmp/prof_replay -name "*.db" | head -1)
python probes/dump_step_trace.py $db $OUT/replay_trace.tsv
wc -l $OUT/replay_trace.tsv; gzip -f $OUT/replay_trace.tsv; ls -la $OUT
