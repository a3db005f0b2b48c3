#!/bin/bash
# A/B by kernel trace: the tree (LEAN dK/dV) against the same tree with LEAN off (probes/_ab/libx2vlm_hip_base.so)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03z
timeout 300 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "attention or attn" > gpurun_out/r03z/kernel_tests.log 2>&1; echo "kernel tests rc=$?"; tail -n 2 gpurun_out/r03z/kernel_tests.log
OLD=$GRAFT_REPO_ROOT/probes/_ab/libx2vlm_hip_base.so
for v in new old; do
  rm -rf /tmp/prof_$v
  if [ $v = old ]; then export X2VLM_HIP_LIB=$OLD; else unset X2VLM_HIP_LIB; fi
  (cd /tmp && rocprofv3 --kernel-trace --stats -d /tmp/prof_$v -o $v -- python $GRAFT_REPO_ROOT/bench.py --config base --serialize --no-graph --steps 3 --warmup 1 --no-cpu-baseline --no-other-configs > $GRAFT_REPO_ROOT/gpurun_out/r03z/prof_$v.log 2>&1)
  db=$(find /tmp/prof_$v -name "*.db" | head -1)
  python probes/prof_summary.py $db 8 > gpurun_out/r03z/${v}_serialized_kernel_stats.txt
  echo "== $v"; head -2 gpurun_out/r03z/${v}_serialized_kernel_stats.txt | tail -1; grep -E "attn_bwd_dkv" gpurun_out/r03z/${v}_serialized_kernel_stats.txt | cut -c1-140
done
unset X2VLM_HIP_LIB
run() { name=$1; cfg=$2; shift; shift; env "$@" timeout 400 python bench.py --config $cfg --steps 16 --warmup 4 --no-cpu-baseline --no-other-configs > gpurun_out/r03z/bench_$name.json 2> gpurun_out/r03z/bench_$name.err; echo -n "$name: "; python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/r03z/bench_$name.json").read().strip().splitlines()[-1])
    print(d["value"], d["ms_per_step"], d["ms_per_step_spread"]["min"], d["ms_per_step_spread"]["median"], d["ms_per_step_spread"]["max"])
except Exception as e: print("ERR", e)
PY
}
run base_new base X2_DUMMY=1
run base_old base X2VLM_HIP_LIB=$OLD
