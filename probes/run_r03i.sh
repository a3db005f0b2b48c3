#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03i
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -m gpu -k "attention" > gpurun_out/r03i/pytest_attn.log 2>&1; echo "rc=$?" >> gpurun_out/r03i/pytest_attn.log
tail -n 4 gpurun_out/r03i/pytest_attn.log | cut -c1-300
timeout 600 python probes/bench_attn.py > gpurun_out/r03i/bench_attn.log 2>&1; head -12 gpurun_out/r03i/bench_attn.log
run() { name=$1; cfg=$2; shift; shift; env "$@" timeout 400 python bench.py --config $cfg --steps 12 --warmup 3 --no-cpu-baseline --no-other-configs > gpurun_out/r03i/bench_$name.json 2> gpurun_out/r03i/bench_$name.err; echo -n "$name: "; python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/r03i/bench_$name.json").read().strip().splitlines()[-1])
    print(d["ms_per_step"], d["ms_per_step_spread"]["min"], d["ms_per_step_spread"]["median"], d["ms_per_step_spread"]["max"], d["launch_mode"])
except Exception as e: print("ERR", e)
PY
}
run large_xcd large X2_DUMMY=1
run large_3d large X2_ATTN_VARIANT=28672
run base_xcd base X2_DUMMY=1
run base_3d base X2_ATTN_VARIANT=28672
run large_xcd2 large X2_DUMMY=1
