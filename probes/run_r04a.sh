#!/bin/bash
# round 3 closing evidence on the final tree: full GPU suite, smoke, the driver's bench command, base serialized kernel trace, HBM traffic PMC passes
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04a
timeout 1500 python -m pytest tests -q -m gpu --maxfail=8 > gpurun_out/r04a/pytest_all.log 2>&1; echo "rc=$?" >> gpurun_out/r04a/pytest_all.log
tail -n 4 gpurun_out/r04a/pytest_all.log | cut -c1-300
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r04a/smoke.log 2>&1; echo "smoke rc=$?"; tail -n 1 gpurun_out/r04a/smoke.log | cut -c1-300
T0=$(date +%s)
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r04a/bench_driver_cmd.json 2> gpurun_out/r04a/bench_driver_cmd.err
echo "driver command wall: $(( $(date +%s) - T0 )) s"
python - <<PY
import json
d=json.loads(open("gpurun_out/r04a/bench_driver_cmd.json").read().strip().splitlines()[-1])
print("base", d["value"], d["ms_per_step"], d["ms_per_step_spread"], d["launch_mode"], "host", d["host_enqueue_ms_per_step"], "frac", d["roofline"]["frac"], d["roofline"]["avg_launch_us"], "step frac", d["roofline"]["also"]["whole_step_frac"], "cpu", d.get("cpu_baseline",{}).get("value"))
for k,v in d.get("other_configs",{}).items(): print(k, {kk: v.get(kk) for kk in ("value","ms_per_step","launch_mode","whole_step_frac","whole_step_tflops","error")}, (v.get("cpu_baseline") or {}).get("value"), v.get("gemm_nt_isolated"))
PY
bash probes/run_prof.sh r04a base > gpurun_out/r04a/prof.log 2>&1
head -12 gpurun_out/r04a/base_serialized_kernel_stats.txt | cut -c1-140
bash probes/run_pmc.sh r04a > gpurun_out/r04a/pmc.log 2>&1; tail -n 3 gpurun_out/r04a/pmc.log | cut -c1-300
