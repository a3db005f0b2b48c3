#!/bin/bash
# runtime knob: kernel arguments in device memory (HIP_FORCE_DEV_KERNARG) under hipGraph-segment replay
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04e
run() { name=$1; cfg=$2; shift; shift; env "$@" timeout 400 python bench.py --config $cfg --steps 16 --warmup 4 --no-cpu-baseline --no-other-configs > gpurun_out/r04e/bench_$name.json 2> gpurun_out/r04e/bench_$name.err; echo -n "$name: "; python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/r04e/bench_$name.json").read().strip().splitlines()[-1])
    print(d["value"], d["ms_per_step"], d["ms_per_step_spread"]["min"], d["ms_per_step_spread"]["median"], d["ms_per_step_spread"]["max"], d["host_enqueue_ms_per_step"])
except Exception as e: print("ERR", e)
PY
}
run base_default base X2_DUMMY=1
run base_devkernarg1 base HIP_FORCE_DEV_KERNARG=1
run base_devkernarg0 base HIP_FORCE_DEV_KERNARG=0
run base_default2 base X2_DUMMY=1
