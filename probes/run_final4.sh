mkdir -p gpurun_out/final4
timeout 1200 python -m pytest tests -m gpu -q > gpurun_out/final4/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/final4/pytest.log
grep -E "passed|failed|rc=" gpurun_out/final4/pytest.log | tail -2
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
for c in base large video; do
timeout 600 python bench.py --config $c --steps 20 --warmup 5 > gpurun_out/final4/bench_$c.log 2>gpurun_out/final4/bench_$c.err
grep '^{' gpurun_out/final4/bench_$c.log | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$c', d['value'], d['unit'], d['ms_per_step'], d['ms_per_step_spread'], d['launch_mode'], 'host', d['host_enqueue_ms_per_step'], 'iso', d['roofline']['achieved'], d['roofline']['frac'], d['roofline']['avg_launch_us'], 'whole', d['roofline']['also']['whole_step_tflops'], d['roofline']['also']['whole_step_frac'], d['cpu_baseline']['value'])"
done
