#!/bin/bash
set -x
OUT=gpurun_out/r5c3
mkdir -p $OUT
timeout 300 python probes/parity_layer_probe.py tiny_text base_shallow_text > $OUT/layer_probe.txt 2>&1
timeout 600 python -m pytest tests/test_graph_gpu.py -x -q -k "text_only or text_part or mixed" > $OUT/graph_tests.log 2>&1
echo "rc graph $?" >> $OUT/summary.txt
NCCL_MAX_NCHANNELS=16 timeout 600 python -m pytest tests/test_ddp_gpu.py -x -q -k "single_rank" > $OUT/rccl_tests.log 2>&1
echo "rc rccl single rank with NCCL_MAX_NCHANNELS=16 $?" >> $OUT/summary.txt
X2_SEG_TIMES=1 timeout 300 python bench.py --steps 20 --warmup 5 --no-other-configs --no-cpu-baseline > $OUT/bench_segtimes.json 2> $OUT/bench_segtimes.err
grep -v amdgpu $OUT/layer_probe.txt | tail -60; cat $OUT/summary.txt; tail -3 $OUT/graph_tests.log; grep "segment times" $OUT/bench_segtimes.err
