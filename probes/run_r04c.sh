#!/bin/bash
# N = 197 dK/dV kernel compiled for 2 waves per SIMD (X2_ATTN_VARIANT bit 15) against the default (4 waves per SIMD, 128-VGPR cap)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04c
run() { name=$1; cfg=$2; shift; shift; env "$@" timeout 400 python bench.py --config $cfg --steps 16 --warmup 4 --no-cpu-baseline --no-other-configs > gpurun_out/r04c/bench_$name.json 2> gpurun_out/r04c/bench_$name.err; echo -n "$name: "; python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/r04c/bench_$name.json").read().strip().splitlines()[-1])
    print(d["value"], d["ms_per_step"], d["ms_per_step_spread"]["min"], d["ms_per_step_spread"]["median"], d["ms_per_step_spread"]["max"])
except Exception as e: print("ERR", e)
PY
}
run base_default base X2_DUMMY=1
run base_2wps base X2_ATTN_VARIANT=45056
run base_default2 base X2_DUMMY=1
run base_2wps2 base X2_ATTN_VARIANT=45056
rm -rf /tmp/prof_v
(cd /tmp && X2_ATTN_VARIANT=45056 rocprofv3 --kernel-trace --stats -d /tmp/prof_v -o v -- python $GRAFT_REPO_ROOT/bench.py --config base --serialize --no-graph --steps 3 --warmup 1 --no-cpu-baseline --no-other-configs > $GRAFT_REPO_ROOT/gpurun_out/r04c/prof_v.log 2>&1)
db=$(find /tmp/prof_v -name "*.db" | head -1)
python probes/prof_summary.py $db 8 > gpurun_out/r04c/2wps_serialized_kernel_stats.txt
head -2 gpurun_out/r04c/2wps_serialized_kernel_stats.txt | tail -1; grep -E "attn_" gpurun_out/r04c/2wps_serialized_kernel_stats.txt | cut -c1-140
