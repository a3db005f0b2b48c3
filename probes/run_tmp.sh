#!/bin/bash
tag=r07c
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/$tag
timeout 200 python probes/bench_rowwise.py 2>&1 | grep -v amdgpu.ids | head -12 > gpurun_out/$tag/rowwise_base.txt
X2_HACK_LNB_PRE=1 timeout 200 python probes/bench_rowwise.py 2>&1 | grep -v amdgpu.ids | head -12 > gpurun_out/$tag/rowwise_pre.txt
paste -d'\n' gpurun_out/$tag/rowwise_base.txt gpurun_out/$tag/rowwise_pre.txt | grep "layernorm_bwd\|stage"
for rep in 1 2; do
  for t in base pre; do
    [ $t = pre ] && export X2_HACK_LNB_PRE=1 || unset X2_HACK_LNB_PRE
    timeout 300 python bench.py --steps 16 --warmup 4 --no-cpu-baseline --no-other-configs 2>/dev/null | tail -1 > gpurun_out/$tag/bench_${t}_$rep.json
    python -c "
import json
d=json.loads(open('gpurun_out/$tag/bench_${t}_$rep.json').read()); print('$t $rep: %.2f ms/step %s' % (d['ms_per_step'], d['ms_per_step_spread']))"
  done
done
