#!/bin/bash
# round 4, call M: upper bound of what moving the fusion layers' weight gradients out of the end phase can give (timing only: they are skipped)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05m
run() { name=$1; shift; env "$@" timeout 400 python bench.py --config base --steps 16 --warmup 4 --no-cpu-baseline --no-other-configs > gpurun_out/r05m/bench_$name.json 2> gpurun_out/r05m/bench_$name.err; echo -n "$name: "; python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/r05m/bench_$name.json").read().strip().splitlines()[-1])
    print(d["value"], d["ms_per_step"], d["ms_per_step_spread"]["min"], d["ms_per_step_spread"]["median"], d["ms_per_step_spread"]["max"])
except Exception as e: print("ERR", e)
PY
grep "segment times" gpurun_out/r05m/bench_$name.err | cut -c1-700
}
run default X2_SEG_TIMES=1
run skipfw X2_SEG_TIMES=1 X2_HACK_SKIP_FW=1
run default2 X2_DUMMY=1
run skipfw2 X2_HACK_SKIP_FW=1
