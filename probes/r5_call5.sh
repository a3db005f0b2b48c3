#!/bin/bash
set -x
OUT=gpurun_out/r5c5
mkdir -p $OUT
export X2_PARITY_DUMP=$OUT/parity X2_PARITY_NO_ASSERT=1
timeout 900 python -m pytest tests/test_layerwise_gpu.py -x -q -s > $OUT/layerwise.log 2>&1
echo "rc layerwise $?" >> $OUT/summary.txt
unset X2_PARITY_DUMP X2_PARITY_NO_ASSERT
timeout 600 python -m pytest tests/test_graph_gpu.py -x -q > $OUT/graph_tests.log 2>&1
echo "rc graph $?" >> $OUT/summary.txt
cat $OUT/summary.txt; grep -v amdgpu $OUT/layerwise.log | grep -E "^\[|^   |passed|failed|Error|error" | head -80; tail -3 $OUT/graph_tests.log
