mkdir -p gpurun_out/r2m
timeout 600 python -m pytest tests/test_kernels_gpu.py -x -q -k "gemm" > gpurun_out/r2m/pytest.log 2>&1; tail -2 gpurun_out/r2m/pytest.log
for c in large base; do
timeout 400 python bench.py --config $c --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/r2m/bench_$c.log 2>&1
python - <<PY
import json
l=[x for x in open('gpurun_out/r2m/bench_$c.log') if x.startswith('{')]
d=json.loads(l[-1]); print('$c', d['value'], d['unit'], d['ms_per_step'], d['ms_per_step_spread'], 'iso', d['roofline']['frac'], 'whole', d['roofline']['also']['whole_step_frac'])
PY
done
