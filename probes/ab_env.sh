#!/bin/bash
# Runtime knobs of the HIP runtime (launch path of the replayed segments), base step: one bench.py run per setting.
# usage: probes/ab_env.sh OUTDIR
out=${1:-gpurun_out/ab_env}; mkdir -p $out
run() { name=$1; shift
  env "$@" X2_BENCH_UNPATCHED=0 X2_GRAPH_CANARY=0 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-other-configs > $out/$name.json 2> $out/$name.err
  python - "$name" $out/$name.json <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[2]) if l.startswith("{")][-1])
    print("%-28s ms_per_step %.3f  spread %s  host %.2f  %s" % (sys.argv[1], d["ms_per_step"], d["ms_per_step_spread"]["median"], d["host_enqueue_ms_per_step"], d["launch_mode"]))
except Exception as e:
    print("%-28s FAILED %s" % (sys.argv[1], e))
PY
}
run default X2_NOP=1
run dev_kernarg1 HIP_FORCE_DEV_KERNARG=1
run dev_kernarg0 HIP_FORCE_DEV_KERNARG=0
run packet_capture1 DEBUG_CLR_GRAPH_PACKET_CAPTURE=1
run packet_capture0 DEBUG_CLR_GRAPH_PACKET_CAPTURE=0
run opt_flush0 AMD_OPT_FLUSH=0
run default2 X2_NOP=1
