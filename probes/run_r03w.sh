#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03w
timeout 600 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "attention or attn" > gpurun_out/r03w/kernel_tests.log 2>&1; echo "kernel tests rc=$?"; tail -n 2 gpurun_out/r03w/kernel_tests.log
run() { name=$1; cfg=$2; shift; shift; env "$@" timeout 400 python bench.py --config $cfg --steps 16 --warmup 4 --no-cpu-baseline --no-other-configs > gpurun_out/r03w/bench_$name.json 2> gpurun_out/r03w/bench_$name.err; echo -n "$name: "; python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/r03w/bench_$name.json").read().strip().splitlines()[-1])
    print(d["value"], d["ms_per_step"], d["ms_per_step_spread"]["min"], d["ms_per_step_spread"]["median"], d["ms_per_step_spread"]["max"], "nt_frac", d["roofline"]["frac"], d["roofline"]["achieved"])
except Exception as e: print("ERR", e)
PY
}
OLD=$GRAFT_REPO_ROOT/probes/_ab/libx2vlm_hip_base.so
run base_new base X2_DUMMY=1
run base_old base X2VLM_HIP_LIB=$OLD
run base_new2 base X2_DUMMY=1
run base_old2 base X2VLM_HIP_LIB=$OLD
python probes/bench_attn.py > gpurun_out/r03w/bench_attn_new.txt 2>&1; X2VLM_HIP_LIB=$OLD python probes/bench_attn.py > gpurun_out/r03w/bench_attn_old.txt 2>&1
grep -E "^vision|^text|^fusion|^cross" gpurun_out/r03w/bench_attn_new.txt | grep -v round; echo ---; grep -E "^vision|^text|^fusion|^cross" gpurun_out/r03w/bench_attn_old.txt | grep -v round
