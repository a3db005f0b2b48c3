#!/bin/bash
# round 4, call L: the N > 1 code path at FULL depth on one GPU: quarters cut at N = 1, and two ranks over gloo (bench.py's own flow, base model, batch 16)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05l
X2_SEG_VISION_CUT=3,6,9 timeout 400 python bench.py --config base --steps 8 --warmup 3 --no-cpu-baseline --no-other-configs > gpurun_out/r05l/bench_quarters.json 2> gpurun_out/r05l/bench_quarters.err
python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/r05l/bench_quarters.json").read().strip().splitlines()[-1]); print("quarters N=1:", d["value"], d["ms_per_step"], d["launch_mode"])
except Exception as e: print("ERR", e); print(open("gpurun_out/r05l/bench_quarters.err").read()[-2000:])
PY
X2_BENCH_BACKEND=gloo X2_DIST_BACKEND=gloo timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 2 --config base --batch 16 --steps 6 --warmup 2 --no-cpu-baseline > gpurun_out/r05l/bench_2rank_gloo.json 2> gpurun_out/r05l/bench_2rank_gloo.err
python - <<PY
import json
try:
    d=json.loads([l for l in open("gpurun_out/r05l/bench_2rank_gloo.json").read().strip().splitlines() if l.startswith("{")][-1]); print("2 ranks gloo, base depth, B=16:", d["value"], d["ms_per_step"], d["launch_mode"], d["n_gpus"], d["config"]["global_batch"])
except Exception as e: print("ERR", e); print(open("gpurun_out/r05l/bench_2rank_gloo.err").read()[-3000:])
PY
for w in 4; do
X2_BENCH_BACKEND=gloo X2_DIST_BACKEND=gloo timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node $w --master-addr 127.0.0.1 --master-port 29612 bench.py --gpus $w --config base --batch 8 --steps 4 --warmup 2 --no-cpu-baseline > gpurun_out/r05l/bench_${w}rank_gloo.json 2> gpurun_out/r05l/bench_${w}rank_gloo.err
python - <<PY
import json
try:
    d=json.loads([l for l in open("gpurun_out/r05l/bench_${w}rank_gloo.json").read().strip().splitlines() if l.startswith("{")][-1]); print("$w ranks gloo, base depth, B=8:", d["value"], d["ms_per_step"], d["launch_mode"], d["n_gpus"])
except Exception as e: print("ERR", e); print(open("gpurun_out/r05l/bench_${w}rank_gloo.err").read()[-3000:])
PY
done
