#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03x
timeout 300 python probes/bench_nt_stagger.py > gpurun_out/r03x/nt_stagger.txt 2>&1; tail -n 34 gpurun_out/r03x/nt_stagger.txt
run() { name=$1; cfg=$2; shift; shift; env "$@" timeout 400 python bench.py --config $cfg --steps 16 --warmup 4 --no-cpu-baseline --no-other-configs > gpurun_out/r03x/bench_$name.json 2> gpurun_out/r03x/bench_$name.err; echo -n "$name: "; python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/r03x/bench_$name.json").read().strip().splitlines()[-1])
    print(d["value"], d["ms_per_step"], d["ms_per_step_spread"]["min"], d["ms_per_step_spread"]["median"], d["ms_per_step_spread"]["max"], "nt_frac", d["roofline"]["frac"], d["roofline"]["achieved"])
except Exception as e: print("ERR", e)
PY
}
run base_off base X2_DUMMY=1
run base_ph2 base X2_TUNE=4=258
run base_ph4 base X2_TUNE=4=260
run base_off2 base X2_DUMMY=1
run base_slot1 base X2_TUNE=4=1
