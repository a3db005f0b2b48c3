#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03t
timeout 500 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "attention or attn" > gpurun_out/r03t/attn_tests.log 2>&1; echo "attn tests rc=$?"; tail -n 3 gpurun_out/r03t/attn_tests.log
timeout 500 python -m pytest tests/test_model_gpu.py -x -q -m gpu > gpurun_out/r03t/model_tests.log 2>&1; echo "model tests rc=$?"; tail -n 3 gpurun_out/r03t/model_tests.log
run() { name=$1; cfg=$2; shift; shift; env "$@" timeout 400 python bench.py --config $cfg --steps 16 --warmup 4 --no-cpu-baseline --no-other-configs > gpurun_out/r03t/bench_$name.json 2> gpurun_out/r03t/bench_$name.err; echo -n "$name: "; python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/r03t/bench_$name.json").read().strip().splitlines()[-1])
    print(d["value"], d["ms_per_step"], d["ms_per_step_spread"]["min"], d["ms_per_step_spread"]["median"], d["ms_per_step_spread"]["max"])
except Exception as e: print("ERR", e)
PY
}
run base_rounds base X2_DUMMY=1
run base_stream base X2_ATTN_VARIANT=77824
run base_rounds2 base X2_DUMMY=1
run base_stream2 base X2_ATTN_VARIANT=77824
run region_rounds region X2_DUMMY=1
run region_stream region X2_ATTN_VARIANT=77824
cd /tmp && rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r03t/prof -o base -- python $GRAFT_REPO_ROOT/bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-other-configs > /dev/null 2>&1
cd $GRAFT_REPO_ROOT; f=$(ls gpurun_out/r03t/prof/*kernel_stats.csv 2>/dev/null | head -1); [ -n "$f" ] && grep -E "attn_bwd_dkv" $f | cut -c1-220
find gpurun_out/r03t/prof -name "*.db" -delete; find gpurun_out/r03t/prof -name "*kernel_trace.csv" -delete
