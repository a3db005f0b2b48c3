# One box: the full GPU suite on the new defaults, a model-level subset on the fall-back switches, then an interleaved
# A/B of this commit's four changes through their switches:
#   X2_TUNE=7=1                 no 160x128 NT tiles (192x128 as before)
#   X2_SPLIT_DECODER_DGRAD=0    decoder input gradient as one un-split NT launch
#   X2_FUSED_MLM_CE=0           fp32 logits + x2_ce_fwd / x2_ce_bwd
#   X2_TAIL_WGRAD=0             last two vision layers' weight gradients as one paired launch
mkdir -p gpurun_out/ab3
timeout 900 python -m pytest tests -m gpu -q > gpurun_out/ab3/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/ab3/pytest.log
grep -E "passed|failed|rc=|^FAILED|^ERROR" gpurun_out/ab3/pytest.log | tail -12
X2_FUSED_MLM_CE=0 X2_TAIL_WGRAD=0 timeout 300 python -m pytest tests/test_model_gpu.py -m gpu -q -k "tiny or base_shallow or base_full_b64" > gpurun_out/ab3/pytest_fallback.log 2>&1
echo "fallback: $(grep -E 'passed|failed' gpurun_out/ab3/pytest_fallback.log | tail -1)"
OFF="X2_TUNE=7=1 X2_SPLIT_DECODER_DGRAD=0 X2_FUSED_MLM_CE=0 X2_TAIL_WGRAD=0"
one() {  # label, env...
  lab=$1; shift
  env "$@" X2_GRAPH_CANARY=0 timeout 300 python bench.py --config ${CFG:-base} --steps 15 --warmup 3 --no-cpu-baseline 2>gpurun_out/ab3/err_$lab.log | tee gpurun_out/ab3/bench_$lab.json | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$lab', d['ms_per_step'], d['ms_per_step_spread']['median'], 'iso', d['roofline']['frac'], d['roofline']['avg_launch_us'], d['launch_mode'])"
}
one off1 $OFF
one on1 X2_TUNE=
one on_no160 X2_TUNE=7=1
one on_notail X2_TUNE= X2_TAIL_WGRAD=0
one on_nofused X2_TUNE= X2_FUSED_MLM_CE=0
one off2 $OFF
one on2 X2_TUNE=
CFG=video one video_off $OFF
CFG=video one video_on X2_TUNE=
CFG=large one large_on X2_TUNE=
