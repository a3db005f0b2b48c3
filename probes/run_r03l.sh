#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03l
for c in base large; do
X2_SEG_TIMES=1 timeout 400 python bench.py --config $c --steps 16 --warmup 4 --no-cpu-baseline --no-other-configs > gpurun_out/r03l/bench_$c.json 2> gpurun_out/r03l/bench_$c.err
grep "segment times" gpurun_out/r03l/bench_$c.err
done
