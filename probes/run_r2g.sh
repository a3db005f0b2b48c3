mkdir -p gpurun_out/r2g
timeout 600 python -m pytest tests/test_kernels_gpu.py -x -q -k "tn" > gpurun_out/r2g/pytest.log 2>&1; tail -5 gpurun_out/r2g/pytest.log
timeout 400 python bench.py --config large --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/r2g/bench_large.log 2>&1
python - <<PY
import json
l=[x for x in open('gpurun_out/r2g/bench_large.log') if x.startswith('{')]
d=json.loads(l[-1]); print('large', d['value'], d['ms_per_step'], 'host', d['host_enqueue_ms_per_step'], d['launch_mode'], d['roofline']['also'])
PY
