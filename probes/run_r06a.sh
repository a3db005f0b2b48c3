#!/bin/bash
# round 4, closing call A: the driver's command (one JSON line incl. other_configs), then serialized kernel traces of base / large / video
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06a
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r06a/bench_driver_cmd.json 2> gpurun_out/r06a/bench_driver_cmd.err
python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/r06a/bench_driver_cmd.json").read().strip().splitlines()[-1])
    print("base", d["value"], d["ms_per_step"], d["ms_per_step_spread"], d["roofline"]["frac"], d["roofline"]["avg_launch_us"], d.get("cpu_baseline"))
    for k,v in d.get("other_configs",{}).items(): print(k, v.get("value"), v.get("ms_per_step"), v.get("whole_step_frac"), v.get("cpu_baseline",{}).get("value") if isinstance(v.get("cpu_baseline"),dict) else v.get("error"))
except Exception as e: print("ERR", e); print(open("gpurun_out/r06a/bench_driver_cmd.err").read()[-2000:])
PY
bash probes/run_prof.sh r06a base large video > gpurun_out/r06a/prof_driver.log 2>&1
head -12 gpurun_out/r06a/base_serialized_kernel_stats.txt | cut -c1-150
