#!/bin/bash
set -x
OUT=gpurun_out/r5c4
mkdir -p $OUT
timeout 600 python -m pytest tests/test_graph_gpu.py -x -q > $OUT/graph_tests.log 2>&1
echo "rc graph $?" >> $OUT/summary.txt
timeout 900 python -m pytest tests/test_ddp_gpu.py -x -q -k "two_ranks_replayed" > $OUT/ddp_tests.log 2>&1
echo "rc ddp two ranks replayed $?" >> $OUT/summary.txt
timeout 900 python probes/ab_step.py --config base --variants "one_chain:X2_TAIL_CHAINS=1" "two_chains:X2_TAIL_CHAINS=2" --rounds 3 --steps 20 > $OUT/ab_tail_chains_base.txt 2>&1
timeout 900 python probes/ab_step.py --config large --variants "one_chain:X2_TAIL_CHAINS=1" "two_chains:X2_TAIL_CHAINS=2" --rounds 2 --steps 10 > $OUT/ab_tail_chains_large.txt 2>&1
X2_SEG_TIMES=1 timeout 300 python bench.py --steps 20 --warmup 5 --no-other-configs --no-cpu-baseline > $OUT/bench_two_chains.json 2> $OUT/bench_two_chains.err
cat $OUT/summary.txt; tail -4 $OUT/graph_tests.log; tail -4 $OUT/ddp_tests.log; grep -v amdgpu $OUT/ab_tail_chains_base.txt | tail -4; grep -v amdgpu $OUT/ab_tail_chains_large.txt | tail -4; grep "segment times" $OUT/bench_two_chains.err; cut -c1-400 $OUT/bench_two_chains.json
