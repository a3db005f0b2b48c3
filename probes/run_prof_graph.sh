mkdir -p gpurun_out/profg
export TMPDIR=/tmp
rm -rf /tmp/prof_g
(cd /tmp && X2_GRAPH_CANARY=0 rocprofv3 --kernel-trace --stats -d /tmp/prof_g -o g -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/profg/bench.log 2>&1)
db=$(find /tmp/prof_g -name "*.db" | head -1)
# steps: 2 eager warm-up + 2 concurrent timed + 1 + 2 serialized + 1 graph warm-up + 1 capture (not executed) + 2 + 5 replays = 15 executed
python probes/prof_summary.py $db 15 > gpurun_out/profg/r02c_kernel_stats.txt
head -12 gpurun_out/profg/r02c_kernel_stats.txt
grep '^{' gpurun_out/profg/bench.log | cut -c1-260
