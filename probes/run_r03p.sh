#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03p
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -m gpu -k "layernorm" > gpurun_out/r03p/pytest_ln.log 2>&1; echo "rc=$?" >> gpurun_out/r03p/pytest_ln.log
tail -n 3 gpurun_out/r03p/pytest_ln.log | cut -c1-300
run() { name=$1; cfg=$2; shift; shift; env "$@" timeout 400 python bench.py --config $cfg --steps 16 --warmup 4 --no-cpu-baseline --no-other-configs > gpurun_out/r03p/bench_$name.json 2> gpurun_out/r03p/bench_$name.err; echo -n "$name: "; python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/r03p/bench_$name.json").read().strip().splitlines()[-1])
    print(d["value"], d["ms_per_step"], d["ms_per_step_spread"]["min"], d["ms_per_step_spread"]["median"], d["ms_per_step_spread"]["max"])
except Exception as e: print("ERR", e)
PY
}
run base_nofuse base X2_FUSE_LAYERSCALE_BWD=0
run base_fuse base X2_FUSE_LAYERSCALE_BWD=1
run base_nofuse2 base X2_FUSE_LAYERSCALE_BWD=0
run base_fuse2 base X2_FUSE_LAYERSCALE_BWD=1
run large_nofuse large X2_FUSE_LAYERSCALE_BWD=0
run large_fuse large X2_FUSE_LAYERSCALE_BWD=1
