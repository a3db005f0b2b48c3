# interleaved A/B of two builds of the library on one box: x2-vlm_amd/libx2vlm_hip_ref.so (A) vs libx2vlm_hip.so (B)
cfg=${1:-base}
for r in 1 2 3; do for v in ref new; do
  lib=$GRAFT_REPO_ROOT/x2-vlm_amd/libx2vlm_hip.so; [ $v = ref ] && lib=$GRAFT_REPO_ROOT/x2-vlm_amd/libx2vlm_hip_ref.so
  X2VLM_HIP_LIB=$lib X2_GRAPH_CANARY=0 timeout 300 python bench.py --config $cfg --steps 15 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$v', d['ms_per_step'], d['ms_per_step_spread']['median'], 'iso', d['roofline']['frac'], d['roofline']['avg_launch_us'])"
done; done
