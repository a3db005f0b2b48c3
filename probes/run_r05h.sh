#!/bin/bash
# round 4, call H: ordered kernel sequence of one serialized base step (where does the tail segment's time go)
cd /tmp && export TMPDIR=/tmp
mkdir -p $GRAFT_REPO_ROOT/gpurun_out/r05h
rm -rf /tmp/prof_seq
rocprofv3 --kernel-trace --stats -d /tmp/prof_seq -o base -- python $GRAFT_REPO_ROOT/bench.py --config base --serialize --no-graph --steps 3 --warmup 1 --no-cpu-baseline --no-other-configs > $GRAFT_REPO_ROOT/gpurun_out/r05h/prof.log 2>&1
db=$(find /tmp/prof_seq -name "*.db" | head -1)
cd $GRAFT_REPO_ROOT
python probes/prof_sequence.py $db > gpurun_out/r05h/base_sequence.txt
python probes/prof_summary.py $db 8 > gpurun_out/r05h/base_serialized_kernel_stats.txt
head -5 gpurun_out/r05h/base_sequence.txt; wc -l gpurun_out/r05h/base_sequence.txt
