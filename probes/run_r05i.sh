#!/bin/bash
# round 4, call I: in-step A/B of two kept-behind-a-knob variants before pruning: grouped cross-attention forward, row-contiguous fp32 epilogue
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05i
run() { name=$1; shift; env "$@" timeout 400 python bench.py --config base --steps 16 --warmup 4 --no-cpu-baseline --no-other-configs > gpurun_out/r05i/bench_$name.json 2> gpurun_out/r05i/bench_$name.err; echo -n "$name: "; python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/r05i/bench_$name.json").read().strip().splitlines()[-1])
    print(d["value"], d["ms_per_step"], d["ms_per_step_spread"]["min"], d["ms_per_step_spread"]["median"], d["ms_per_step_spread"]["max"], d["roofline"]["frac"])
except Exception as e: print("ERR", e)
PY
}
run default X2_DUMMY=1
run grouped_xfwd X2_ATTN_VARIANT=12296
run f4 X2_TUNE=2=64
run default2 X2_DUMMY=1
run grouped_xfwd2 X2_ATTN_VARIANT=12296
run f4b X2_TUNE=2=64
