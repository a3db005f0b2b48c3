#!/bin/bash
set -x
OUT=gpurun_out/r5c24
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_graph_gpu.py tests/test_train_mode_gpu.py tests/test_layerwise_gpu.py -q -m gpu -x > $OUT/pytest_graph.log 2>&1; rc=$?; echo "rc=$rc" >> $OUT/pytest_graph.log
grep -v "UserWarning\|Consider using\|return Variable\|^$\|amdgpu.ids" $OUT/pytest_graph.log | tail -n 12 | cut -c1-400
timeout 600 python probes/ab_step.py --config base --variants "default:" "aux1:X2_AUX_OVERLAP=1" "aux0:X2_AUX_OVERLAP=0" --rounds 3 --steps 20 > $OUT/ab_base.txt 2>&1; tail -5 $OUT/ab_base.txt
timeout 600 python probes/ab_step.py --config large --variants "default:" "aux0:X2_AUX_OVERLAP=0" --rounds 2 --steps 10 > $OUT/ab_large.txt 2>&1; tail -4 $OUT/ab_large.txt
timeout 600 python bench.py --no-cpu-baseline > $OUT/bench.json 2> $OUT/bench.err; python - <<'PY'
import json
for l in open("gpurun_out/r5c24/bench.json"):
    if l.startswith("{"):
        d=json.loads(l); print({k:d[k] for k in ("value","ms_per_step","host_enqueue_ms_per_step","launch_mode")}); print([(c.get("config"),c.get("ms_per_step"),c.get("host_enqueue_ms_per_step")) for c in d.get("other_configs",[])] if isinstance(d.get("other_configs"),list) else d.get("other_configs"))
PY
