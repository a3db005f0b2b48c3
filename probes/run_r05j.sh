#!/bin/bash
# round 4, call J: pruned library (attention bits, w8 / stagger / f4 knob) parity + fused LayerNorm-bwd+layer-scale at a forced 128-VGPR bound (spills 29 dwords) in-step A/B
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05j
timeout 1500 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu > gpurun_out/r05j/pytest_kernels.log 2>&1; echo "rc=$?" >> gpurun_out/r05j/pytest_kernels.log
grep -v "UserWarning\|Consider using\|return Variable\|^$\|amdgpu.ids" gpurun_out/r05j/pytest_kernels.log | tail -n 12 | cut -c1-600
run() { name=$1; shift; env "$@" timeout 400 python bench.py --config base --steps 16 --warmup 4 --no-cpu-baseline --no-other-configs > gpurun_out/r05j/bench_$name.json 2> gpurun_out/r05j/bench_$name.err; echo -n "$name: "; python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/r05j/bench_$name.json").read().strip().splitlines()[-1])
    print(d["value"], d["ms_per_step"], d["ms_per_step_spread"]["min"], d["ms_per_step_spread"]["median"], d["ms_per_step_spread"]["max"], d["roofline"]["frac"])
except Exception as e: print("ERR", e)
PY
}
run default X2_DUMMY=1
run fusedls128 X2_FUSE_LAYERSCALE_BWD=1
run default2 X2_DUMMY=1
run fusedls128b X2_FUSE_LAYERSCALE_BWD=1
