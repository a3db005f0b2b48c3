#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03g
timeout 1800 python -m pytest tests -q -m gpu --maxfail=8 > gpurun_out/r03g/pytest_all.log 2>&1; echo "rc=$?" >> gpurun_out/r03g/pytest_all.log
tail -n 5 gpurun_out/r03g/pytest_all.log | cut -c1-300
/usr/bin/time -v python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r03g/bench_driver_cmd.json 2> gpurun_out/r03g/bench_driver_cmd.err
grep "Elapsed (wall" gpurun_out/r03g/bench_driver_cmd.err
python - <<PY
import json
d=json.loads(open("gpurun_out/r03g/bench_driver_cmd.json").read().strip().splitlines()[-1])
print("base", d["value"], d["ms_per_step"], d["launch_mode"], "host", d["host_enqueue_ms_per_step"], "frac", d["roofline"]["frac"], "step frac", d["roofline"]["also"]["whole_step_frac"], "cpu", d.get("cpu_baseline",{}).get("value"))
for k,v in d.get("other_configs",{}).items(): print(k, {kk: v.get(kk) for kk in ("value","ms_per_step","launch_mode","whole_step_frac","error")}, (v.get("cpu_baseline") or {}).get("value"))
PY
bash probes/run_prof.sh r03g base large video > gpurun_out/r03g/prof.log 2>&1
head -12 gpurun_out/r03g/base_serialized_kernel_stats.txt
bash probes/run_pmc.sh r03g > gpurun_out/r03g/pmc.log 2>&1; tail -n 3 gpurun_out/r03g/pmc.log
