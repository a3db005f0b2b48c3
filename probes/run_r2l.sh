mkdir -p gpurun_out/r2l
timeout 600 python -m pytest tests/test_kernels_gpu.py tests/test_model_gpu.py -x -q > gpurun_out/r2l/pytest.log 2>&1; tail -2 gpurun_out/r2l/pytest.log
bash probes/run_prof.sh r2l base large video > gpurun_out/r2l/prof.log 2>&1
for c in base large video; do head -3 gpurun_out/r2l/${c}_serialized_kernel_stats.txt | tail -1; done
for c in base large video; do
timeout 600 python bench.py --config $c --steps 20 --warmup 5 > gpurun_out/r2l/bench_$c.log 2>&1
python - <<PY
import json
l=[x for x in open('gpurun_out/r2l/bench_$c.log') if x.startswith('{')]
d=json.loads(l[-1]); print('$c', d['value'], d['unit'], d['ms_per_step'], d['ms_per_step_spread'], 'host', d['host_enqueue_ms_per_step'], d['launch_mode'], 'iso', d['roofline']['frac'], 'whole', d['roofline']['also']['whole_step_frac'], d.get('cpu_baseline'))
PY
done
