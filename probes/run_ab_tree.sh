# A/B of two source trees on one box: $GRAFT_REPO_ROOT (new) vs $GRAFT_REPO_ROOT/_ref_tree (a `git archive` of the reference commit + its built library)
cfg=${1:-base}
for r in 1 2 3; do for v in ref new; do
  d=$GRAFT_REPO_ROOT; [ $v = ref ] && d=$GRAFT_REPO_ROOT/_ref_tree
  (cd $d && X2_GRAPH_CANARY=0 timeout 300 python bench.py --config $cfg --steps 15 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$v', d['ms_per_step'], d['ms_per_step_spread']['median'])")
done; done
