#!/bin/bash
# 256-column NT kernel: CU-fill threshold of its automatic rule (x2_tune(9, percent); default 80)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04d
run() { name=$1; cfg=$2; shift; shift; env "$@" timeout 400 python bench.py --config $cfg --steps 16 --warmup 4 --no-cpu-baseline --no-other-configs > gpurun_out/r04d/bench_$name.json 2> gpurun_out/r04d/bench_$name.err; echo -n "$name: "; python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/r04d/bench_$name.json").read().strip().splitlines()[-1])
    print(d["value"], d["ms_per_step"], d["ms_per_step_spread"]["min"], d["ms_per_step_spread"]["median"], d["ms_per_step_spread"]["max"], d["roofline"]["frac"])
except Exception as e: print("ERR", e)
PY
}
run base_80 base X2_DUMMY=1
run base_60 base X2_TUNE=9=60
run base_92 base X2_TUNE=9=92
run base_80b base X2_DUMMY=1
run large_80 large X2_DUMMY=1
run large_60 large X2_TUNE=9=60
