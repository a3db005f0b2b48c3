# usage: bash probes/run_timeline.sh <outdir>  -> <outdir>/timeline.txt : stream timeline of the replayed base step (probes/timeline.py)
out=$1; mkdir -p $out
export TMPDIR=/tmp
rm -rf /tmp/prof_tl
(cd /tmp && X2_GRAPH_CANARY=0 X2_BENCH_UNPATCHED=0 rocprofv3 --kernel-trace -d /tmp/prof_tl -o tl -- python $GRAFT_REPO_ROOT/bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-other-configs > $GRAFT_REPO_ROOT/$out/bench.log 2>&1)
db=$(find /tmp/prof_tl -name "*.db" | head -1)
python probes/timeline.py $db > $out/timeline.txt
