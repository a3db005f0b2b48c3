#!/bin/bash
# round 4, call K: glue kernels (tail index, DropPath rows, frame mean), batched arena fills: parity + ATen launch count of the replayed step
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05k
timeout 1500 python -m pytest tests/test_kernels_gpu.py tests/test_model_gpu.py tests/test_train_mode_gpu.py tests/test_graph_gpu.py -x -q -m gpu -k "tail_index or droppath or frame_mean or tiny or base_shallow or video_full or base_region or train_mode or drop_path or dropout or graph or segmented or mixed" > gpurun_out/r05k/pytest.log 2>&1; echo "rc=$?" >> gpurun_out/r05k/pytest.log
grep -v "UserWarning\|Consider using\|return Variable\|^$\|amdgpu.ids" gpurun_out/r05k/pytest.log | tail -n 14 | cut -c1-700
for cfg in base video; do
  timeout 600 python bench.py --config $cfg --steps 16 --warmup 4 --no-cpu-baseline --no-other-configs > gpurun_out/r05k/bench_$cfg.json 2> gpurun_out/r05k/bench_$cfg.err
  python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/r05k/bench_$cfg.json").read().strip().splitlines()[-1])
    print("$cfg", d["value"], d["ms_per_step"], d["ms_per_step_spread"], d["launch_mode"], d["roofline"]["frac"])
except Exception as e: print("$cfg ERR", e); print(open("gpurun_out/r05k/bench_$cfg.err").read()[-1500:])
PY
done
cd /tmp; rm -rf /tmp/prof_rep
rocprofv3 --kernel-trace -d /tmp/prof_rep -o base -- python $GRAFT_REPO_ROOT/bench.py --config base --steps 4 --warmup 2 --no-cpu-baseline --no-other-configs > $GRAFT_REPO_ROOT/gpurun_out/r05k/prof_replay.log 2>&1
db=$(find /tmp/prof_rep -name "*.db" | head -1)
cd $GRAFT_REPO_ROOT; python probes/aten_count.py $db > gpurun_out/r05k/aten_replayed_step.txt; cat gpurun_out/r05k/aten_replayed_step.txt | cut -c1-160
