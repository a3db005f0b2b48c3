#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03s
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r03s/smoke.log 2>&1; echo "smoke rc=$?"; tail -n 1 gpurun_out/r03s/smoke.log | cut -c1-300
X2_BENCH_BACKEND=gloo timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 4 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 4 --steps 3 --warmup 1 --tiny --batch 4 --no-cpu-baseline > gpurun_out/r03s/bench4.log 2>&1; echo "4-rank rc=$?"; grep '"metric"' gpurun_out/r03s/bench4.log | cut -c1-400
run() { name=$1; cfg=$2; shift; shift; env "$@" timeout 400 python bench.py --config $cfg --steps 16 --warmup 4 --no-cpu-baseline --no-other-configs > gpurun_out/r03s/bench_$name.json 2> gpurun_out/r03s/bench_$name.err; echo -n "$name: "; python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/r03s/bench_$name.json").read().strip().splitlines()[-1])
    print(d["value"], d["ms_per_step"], d["ms_per_step_spread"]["min"], d["ms_per_step_spread"]["median"], d["ms_per_step_spread"]["max"])
except Exception as e: print("ERR", e)
PY
}
run base_default base X2_DUMMY=1
run base_grouped_fwd base X2_ATTN_VARIANT=12296
run base_perrow_dq base X2_ATTN_VARIANT=12304
run base_default2 base X2_DUMMY=1
run base_grouped_fwd2 base X2_ATTN_VARIANT=12296
