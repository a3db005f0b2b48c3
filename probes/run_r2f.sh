mkdir -p gpurun_out/r2f
for cfg in "1 1" "0 1" "1 0" "0 0"; do
set -- $cfg
X2_OVERLAP_TOWERS=$1 X2_SIDE_STREAM=$2 timeout 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/r2f/b_$1$2.log 2>&1
python - <<PY
import json
l=[x for x in open('gpurun_out/r2f/b_$1$2.log') if x.startswith('{')]
if l:
    d=json.loads(l[-1]); print('overlap=$1 side=$2', d['value'], d['ms_per_step'], d['ms_per_step_spread'], 'host', d['host_enqueue_ms_per_step'], d['launch_mode'])
else:
    print('overlap=$1 side=$2 FAILED'); print(open('gpurun_out/r2f/b_$1$2.log').read()[-800:])
PY
done
