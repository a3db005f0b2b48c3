mkdir -p gpurun_out/r2j
timeout 600 python -m pytest tests/test_kernels_gpu.py tests/test_model_gpu.py -x -q > gpurun_out/r2j/pytest.log 2>&1; tail -3 gpurun_out/r2j/pytest.log
for c in base large video; do
timeout 400 python bench.py --config $c --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/r2j/bench_$c.log 2>&1
python - <<PY
import json
l=[x for x in open('gpurun_out/r2j/bench_$c.log') if x.startswith('{')]
d=json.loads(l[-1]); print('$c', d['value'], d['unit'], d['ms_per_step'], d['ms_per_step_spread'], 'host', d['host_enqueue_ms_per_step'], d['launch_mode'], 'iso frac', d['roofline']['frac'], 'whole', d['roofline']['also']['whole_step_frac'])
PY
done
