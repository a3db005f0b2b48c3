#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03r
T0=$(date +%s)
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r03r/bench_driver_cmd.json 2> gpurun_out/r03r/bench_driver_cmd.err
echo "driver command wall: $(( $(date +%s) - T0 )) s"
python - <<PY
import json
d=json.loads(open("gpurun_out/r03r/bench_driver_cmd.json").read().strip().splitlines()[-1])
print("base", d["value"], d["ms_per_step"], d["ms_per_step_spread"], d["launch_mode"], "host", d["host_enqueue_ms_per_step"], "frac", d["roofline"]["frac"], d["roofline"]["avg_launch_us"], "step frac", d["roofline"]["also"]["whole_step_frac"], "cpu", d.get("cpu_baseline",{}).get("value"))
for k,v in d.get("other_configs",{}).items(): print(k, {kk: v.get(kk) for kk in ("value","ms_per_step","launch_mode","whole_step_frac","whole_step_tflops","error")}, (v.get("cpu_baseline") or {}).get("value"), v.get("gemm_nt_isolated"))
PY
