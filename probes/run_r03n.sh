#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03n
timeout 1800 python -m pytest tests -q -m gpu --maxfail=8 > gpurun_out/r03n/pytest_all.log 2>&1; echo "rc=$?" >> gpurun_out/r03n/pytest_all.log
tail -n 4 gpurun_out/r03n/pytest_all.log | cut -c1-300
run() { name=$1; cfg=$2; shift; shift; env "$@" timeout 400 python bench.py --config $cfg --steps 16 --warmup 4 --no-cpu-baseline --no-other-configs > gpurun_out/r03n/bench_$name.json 2> gpurun_out/r03n/bench_$name.err; echo -n "$name: "; python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/r03n/bench_$name.json").read().strip().splitlines()[-1])
    print(d["value"], d["ms_per_step"], d["ms_per_step_spread"]["min"], d["ms_per_step_spread"]["median"], d["ms_per_step_spread"]["max"], "host", d["host_enqueue_ms_per_step"])
except Exception as e: print("ERR", e)
PY
}
run base_prefetch base X2_DUMMY=1
run base_noprefetch base X2_SEG_PREFETCH_CASTS=0
run base_prefetch2 base X2_DUMMY=1
run large_prefetch large X2_DUMMY=1
run large_noprefetch large X2_SEG_PREFETCH_CASTS=0
