python probes/bench_gemm.py 1 2>&1 | grep -v Warn | sed -n 2,16p
timeout 300 python -m pytest tests/test_kernels_gpu.py -q -k gemm 2>&1 | tail -1
X2_GRAPH_CANARY=0 python bench.py --steps 15 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('base', d['ms_per_step'], d['ms_per_step_spread']['median'], 'iso', d['roofline']['frac'])"
