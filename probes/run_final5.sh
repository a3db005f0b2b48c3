# final tree of the round: bench lines first, then the full GPU suite
mkdir -p gpurun_out/final5
timeout 120 python bench.py --config base --steps 20 --warmup 5 > gpurun_out/final5/bench_base.log 2>gpurun_out/final5/bench_base.err
timeout 60 python bench.py --config video --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/final5/bench_video.log 2>gpurun_out/final5/bench_video.err
for c in base video; do grep '^{' gpurun_out/final5/bench_$c.log | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$c', d['value'], d['unit'], d['ms_per_step'], d['ms_per_step_spread'], d['launch_mode'], 'host', d['host_enqueue_ms_per_step'], 'iso', d['roofline']['achieved'], d['roofline']['frac'], d['roofline']['avg_launch_us'], 'whole', d['roofline']['also']['whole_step_tflops'], d['roofline']['also']['whole_step_frac'])"; done
timeout 130 python -m pytest tests -m gpu -q > gpurun_out/final5/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/final5/pytest.log
grep -E "passed|failed|rc=|^FAILED|^ERROR" gpurun_out/final5/pytest.log | tail -6
