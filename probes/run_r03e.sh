#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03e
timeout 1500 python -m pytest tests/test_kernels_gpu.py tests/test_graph_gpu.py tests/test_ddp_gpu.py tests/test_model_gpu.py tests/test_train_mode_gpu.py tests/test_bench_gpu.py -q -m gpu --maxfail=8 -k "multi_tensor or kv_csr or layernorm_bwd_with or dgelu or graph or segment or replay or tiny or base_shallow or bench or dropout" > gpurun_out/r03e/pytest.log 2>&1; echo "rc=$?" >> gpurun_out/r03e/pytest.log
tail -n 6 gpurun_out/r03e/pytest.log | cut -c1-300
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs"
timeout 300 $B > gpurun_out/r03e/bench_default.json 2> gpurun_out/r03e/bench_default.err
X2_SEG_TAIL_WGRAD=0 timeout 300 $B > gpurun_out/r03e/bench_notailq.json 2>/dev/null
timeout 300 $B > gpurun_out/r03e/bench_default2.json 2>/dev/null
for f in default notailq default2; do echo -n "$f: "; python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/r03e/bench_$f.json").read().strip().splitlines()[-1])
    print(d["ms_per_step"], d["ms_per_step_spread"], "host", d["host_enqueue_ms_per_step"], d["launch_mode"])
except Exception as e: print("ERR", e)
PY
done
tail -n 5 gpurun_out/r03e/bench_default.err | cut -c1-300
