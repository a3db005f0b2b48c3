#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03f
timeout 1500 python -m pytest tests/test_graph_gpu.py tests/test_ddp_gpu.py tests/test_model_gpu.py tests/test_train_mode_gpu.py tests/test_bench_gpu.py tests/test_fullsize_properties_gpu.py -q -m gpu --maxfail=8 -k "graph or segment or replay or tiny or base_shallow or bench or dropout or halves or drop_path" > gpurun_out/r03f/pytest.log 2>&1; echo "rc=$?" >> gpurun_out/r03f/pytest.log
tail -n 6 gpurun_out/r03f/pytest.log | cut -c1-300
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs"
run() { name=$1; shift; env "$@" timeout 300 $B > gpurun_out/r03f/bench_$name.json 2> gpurun_out/r03f/bench_$name.err; echo -n "$name: "; python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/r03f/bench_$name.json").read().strip().splitlines()[-1])
    print(d["ms_per_step"], d["ms_per_step_spread"]["min"], d["ms_per_step_spread"]["median"], d["ms_per_step_spread"]["max"], "host", d["host_enqueue_ms_per_step"], d["launch_mode"])
except Exception as e: print("ERR", e)
PY
}
run default X2_DUMMY=1
run nocut X2_SEG_VISION_CUT=
run nocut_notailq X2_SEG_VISION_CUT= X2_SEG_TAIL_WGRAD=0
run cut_novw X2_SEG_VISION_WGRAD=0
run cut_novw_notailq X2_SEG_VISION_WGRAD=0 X2_SEG_TAIL_WGRAD=0
run cut3 X2_SEG_VISION_CUT=4,8
run onestream X2_SEG_ONE_STREAM=1 X2_SEG_VISION_CUT= X2_SEG_TAIL_WGRAD=0
run default2 X2_DUMMY=1
tail -n 3 gpurun_out/r03f/bench_default.err | cut -c1-300
