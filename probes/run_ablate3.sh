for v in 0 4; do
  X2_HACK_NT_ABLATE=$v X2_GRAPH_CANARY=0 timeout 300 python bench.py --config ${1:-base} --steps 15 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('${1:-base} ablate=$v', d['ms_per_step'])"
done
