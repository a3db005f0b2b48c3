#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03m
run() { name=$1; cfg=$2; shift; shift; env "$@" timeout 400 python bench.py --config $cfg --steps 16 --warmup 4 --no-cpu-baseline --no-other-configs > gpurun_out/r03m/bench_$name.json 2> gpurun_out/r03m/bench_$name.err; echo -n "$name: "; python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/r03m/bench_$name.json").read().strip().splitlines()[-1])
    print(d["value"], d["ms_per_step"], d["ms_per_step_spread"]["min"], d["ms_per_step_spread"]["median"], d["ms_per_step_spread"]["max"], "host", d["host_enqueue_ms_per_step"])
except Exception as e: print("ERR", e)
PY
}
run base_default base X2_DUMMY=1
run base_c4 base X2_SEG_VISION_CUT=3,6,9
run base_c6 base X2_SEG_VISION_CUT=2,4,6,8,10
run base_c12 base X2_SEG_VISION_CUT=1,2,3,4,5,6,7,8,9,10,11
run large_default large X2_DUMMY=1
run large_c4 large X2_SEG_VISION_CUT=6,12,18
run large_c6 large X2_SEG_VISION_CUT=4,8,12,16,20
run large_c12 large X2_SEG_VISION_CUT=2,4,6,8,10,12,14,16,18,20,22
run base_default2 base X2_DUMMY=1
