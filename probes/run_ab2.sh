var=$1; a=$2; b=$3; cfg=${4:-base}
for r in 1 2; do for v in $a $b; do
  env $var=$v X2_GRAPH_CANARY=0 timeout 300 python bench.py --config $cfg --steps 15 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$var=$v', d['ms_per_step'], d['ms_per_step_spread']['median'])"
done; done
