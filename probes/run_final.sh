mkdir -p gpurun_out/final
timeout 1200 python -m pytest tests -m gpu -q > gpurun_out/final/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/final/pytest.log
grep -E "passed|failed|rc=" gpurun_out/final/pytest.log | tail -3
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
( time python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/final/bench_default.log 2>gpurun_out/final/bench_default.err ) 2>&1 | grep real
grep '^{' gpurun_out/final/bench_default.log | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['value'], d['unit'], d['ms_per_step'], d['ms_per_step_spread'], d['launch_mode'], 'host', d['host_enqueue_ms_per_step'], 'roof', d['roofline']['frac'], d['roofline']['traffic'], d['cpu_baseline'])"
