#!/bin/bash
# round 4, call N: fusion-stack backward cut per layer, weight gradients of each stage on stream B under the stages below: parity + A/B
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05n
timeout 1500 python -m pytest tests/test_graph_gpu.py tests/test_ddp_gpu.py -x -q -m gpu -k "segmented or mixed or replayed" > gpurun_out/r05n/pytest.log 2>&1; echo "rc=$?" >> gpurun_out/r05n/pytest.log
grep -v "UserWarning\|Consider using\|return Variable\|^$\|amdgpu.ids" gpurun_out/r05n/pytest.log | tail -n 14 | cut -c1-900
run() { name=$1; shift; env "$@" timeout 400 python bench.py --config base --steps 16 --warmup 4 --no-cpu-baseline --no-other-configs > gpurun_out/r05n/bench_$name.json 2> gpurun_out/r05n/bench_$name.err; echo -n "$name: "; python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/r05n/bench_$name.json").read().strip().splitlines()[-1])
    print(d["value"], d["ms_per_step"], d["ms_per_step_spread"]["min"], d["ms_per_step_spread"]["median"], d["ms_per_step_spread"]["max"], d["host_enqueue_ms_per_step"], d["launch_mode"])
except Exception as e: print("ERR", e); print(open("gpurun_out/r05n/bench_$name.err").read()[-1500:])
PY
grep "segment times" gpurun_out/r05n/bench_$name.err | cut -c1-900
}
run nocut X2_SEG_FUSION_CUT=
run cut_all X2_SEG_TIMES=1
run cut_half X2_SEG_FUSION_CUT=15
run cut_2 X2_SEG_FUSION_CUT=14,16
run nocut2 X2_SEG_FUSION_CUT=
run cut_all2 X2_DUMMY=1
