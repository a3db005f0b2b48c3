#!/bin/bash
# round 4, call E: whole GPU suite on the current tree + bench lines (base, mixed)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05e
timeout 3000 python -m pytest tests -x -q -m gpu > gpurun_out/r05e/pytest_gpu.log 2>&1; echo "rc=$?" >> gpurun_out/r05e/pytest_gpu.log
grep -v "UserWarning\|Consider using\|return Variable\|^$\|amdgpu.ids" gpurun_out/r05e/pytest_gpu.log | tail -n 40 | cut -c1-800
for cfg in base mixed; do
  timeout 600 python bench.py --config $cfg --steps 16 --warmup 4 --no-cpu-baseline --no-other-configs > gpurun_out/r05e/bench_$cfg.json 2> gpurun_out/r05e/bench_$cfg.err
  python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/r05e/bench_$cfg.json").read().strip().splitlines()[-1])
    print("$cfg", d["value"], d["ms_per_step"], d["ms_per_step_spread"], d["launch_mode"], d["roofline"]["frac"], d["config"]["losses"])
except Exception as e: print("$cfg ERR", e); print(open("gpurun_out/r05e/bench_$cfg.err").read()[-1500:])
PY
done
