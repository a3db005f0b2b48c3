#!/bin/bash
set -x
OUT=gpurun_out/r5c25
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_model_gpu.py tests/test_graph_gpu.py tests/test_retrieval.py -q -m gpu -x -k "region or tiny or graph or mixed or retrieval or rerank" > $OUT/pytest.log 2>&1; rc=$?; echo "rc=$rc" >> $OUT/pytest.log
grep -v "UserWarning\|Consider using\|return Variable\|^$\|amdgpu.ids" $OUT/pytest.log | tail -n 12 | cut -c1-400
timeout 600 python bench.py --no-cpu-baseline > $OUT/bench.json 2> $OUT/bench.err; python - <<'PY'
import json
for l in open("gpurun_out/r5c25/bench.json"):
    if l.startswith("{"):
        d=json.loads(l); print({k:d[k] for k in ("value","ms_per_step","host_enqueue_ms_per_step","launch_mode")})
        for k,c in d.get("other_configs",{}).items(): print(k, c.get("ms_per_step"), c.get("host_enqueue_ms_per_step"))
PY
