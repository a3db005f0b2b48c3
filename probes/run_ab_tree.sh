# A/B of two source trees on one box: the snapshot at $GRAFT_REPO_ROOT (B) vs the same with x2-vlm_amd/engine.py from engine_ref.py and the ref library (A)
cfg=${1:-base}
cp -r $GRAFT_REPO_ROOT /tmp/tree_ref; cp /tmp/tree_ref/probes/engine_ref.py "/tmp/tree_ref/x2-vlm_amd/engine.py"; cp "/tmp/tree_ref/x2-vlm_amd/libx2vlm_hip_ref.so" "/tmp/tree_ref/x2-vlm_amd/libx2vlm_hip.so"
for r in 1 2 3; do for v in ref new; do
  d=$GRAFT_REPO_ROOT; [ $v = ref ] && d=/tmp/tree_ref
  (cd $d && X2_GRAPH_CANARY=0 timeout 300 python bench.py --config $cfg --steps 15 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$v', d['ms_per_step'], d['ms_per_step_spread']['median'])")
done; done
