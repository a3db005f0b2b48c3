#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03h
T0=$(date +%s)
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r03h/bench_driver_cmd.json 2> gpurun_out/r03h/bench_driver_cmd.err
echo "driver command wall: $(( $(date +%s) - T0 )) s"
python - <<PY
import json
d=json.loads(open("gpurun_out/r03h/bench_driver_cmd.json").read().strip().splitlines()[-1])
print("base", d["value"], d["ms_per_step"], d["launch_mode"], "host", d["host_enqueue_ms_per_step"], "frac", d["roofline"]["frac"], "step frac", d["roofline"]["also"]["whole_step_frac"], "cpu", d.get("cpu_baseline",{}).get("value"))
for k,v in d.get("other_configs",{}).items(): print(k, {kk: v.get(kk) for kk in ("value","ms_per_step","launch_mode","whole_step_frac","whole_step_tflops","error")}, (v.get("cpu_baseline") or {}).get("value"))
PY
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs"
run() { name=$1; shift; env "$@" timeout 300 $B > gpurun_out/r03h/bench_$name.json 2> gpurun_out/r03h/bench_$name.err; echo -n "$name: "; python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/r03h/bench_$name.json").read().strip().splitlines()[-1])
    print(d["ms_per_step"], d["ms_per_step_spread"]["min"], d["ms_per_step_spread"]["median"], d["ms_per_step_spread"]["max"], "host", d["host_enqueue_ms_per_step"], d["launch_mode"], "nt_us", d["roofline"]["avg_launch_us"])
except Exception as e: print("ERR", e)
PY
}
run default X2_DUMMY=1
run f4stores X2_TUNE=2=64
run groupm4 X2_TUNE=0=4
run groupm16 X2_TUNE=0=16
run default2 X2_DUMMY=1
