#!/bin/bash
# round 3, call C: full GPU suite on the new kernels (fused LN-bwd/layer-scale, dgelu+colparts, NT256 auto, walk attention default)
# + in-step A/B of the new switches
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03c
timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/r03c/pytest_all.log 2>&1; echo "rc=$?" >> gpurun_out/r03c/pytest_all.log
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline"
timeout 300 $B > gpurun_out/r03c/bench_default.json 2> gpurun_out/r03c/bench_default.err
X2_SEG_SIDE=1 timeout 300 $B > gpurun_out/r03c/bench_segside.json 2> gpurun_out/r03c/bench_segside.err
X2_FUSE_LAYERSCALE_BWD=0 timeout 300 $B > gpurun_out/r03c/bench_nolsfuse.json 2>/dev/null
X2_FUSE_DGELU_COLSUM=0 timeout 300 $B > gpurun_out/r03c/bench_nocolparts.json 2>/dev/null
X2_TUNE=1=1 timeout 300 $B > gpurun_out/r03c/bench_nont256.json 2>/dev/null
X2_ATTN_VARIANT=0 timeout 300 $B > gpurun_out/r03c/bench_nowalk.json 2>/dev/null
timeout 300 $B --graph whole > gpurun_out/r03c/bench_whole.json 2>/dev/null
tail -n 5 gpurun_out/r03c/pytest_all.log | cut -c1-300
for f in default segside nolsfuse nocolparts nont256 nowalk whole; do echo -n "$f: "; python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/r03c/bench_$f.json").read().strip().splitlines()[-1])
    print(d["ms_per_step"], d["ms_per_step_spread"]["median"], "host", d["host_enqueue_ms_per_step"], d["launch_mode"], "nt_iso_us", d["roofline"]["avg_launch_us"], d["roofline"]["frac"])
except Exception as e: print("ERR", e)
PY
done
