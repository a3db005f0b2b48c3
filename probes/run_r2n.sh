mkdir -p gpurun_out/r2n
timeout 900 python -m pytest tests -m gpu -q -x > gpurun_out/r2n/pytest.log 2>&1; grep -E "passed|failed" gpurun_out/r2n/pytest.log | tail -2
for c in base large video; do
timeout 400 python bench.py --config $c --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/r2n/bench_$c.log 2>&1
python - <<PY
import json
l=[x for x in open('gpurun_out/r2n/bench_$c.log') if x.startswith('{')]
d=json.loads(l[-1]); print('$c', d['value'], d['unit'], d['ms_per_step'], d['ms_per_step_spread'], d['launch_mode'], 'iso', d['roofline']['frac'], 'whole', d['roofline']['also']['whole_step_frac'])
PY
done
