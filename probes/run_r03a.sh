#!/bin/bash
# round 3, call A: new tests (walk attention variants, segmented graphs, 2-rank replay, bench flow) + A/B benches
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03a
timeout 900 python -m pytest tests/test_kernels_gpu.py -k attention tests/test_graph_gpu.py -x -q -m gpu > gpurun_out/r03a/pytest1.log 2>&1; echo "pytest1 rc=$?" >> gpurun_out/r03a/pytest1.log
timeout 900 python -m pytest tests/test_ddp_gpu.py tests/test_bench_gpu.py -x -q -m gpu > gpurun_out/r03a/pytest2.log 2>&1; echo "pytest2 rc=$?" >> gpurun_out/r03a/pytest2.log
timeout 300 python probes/bench_attn.py > gpurun_out/r03a/bench_attn.log 2>&1
timeout 400 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r03a/bench_segments.json 2> gpurun_out/r03a/bench_segments.err
timeout 400 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --graph whole > gpurun_out/r03a/bench_whole.json 2> gpurun_out/r03a/bench_whole.err
tail -3 gpurun_out/r03a/pytest1.log gpurun_out/r03a/pytest2.log; cat gpurun_out/r03a/bench_attn.log | head -8; cut -c1-600 gpurun_out/r03a/bench_segments.json gpurun_out/r03a/bench_whole.json
