# usage: bash probes/run_prof.sh <tag> [configs...]   -> gpurun_out/<tag>/<config>_serialized_kernel_stats.txt
tag=$1; shift
mkdir -p gpurun_out/$tag
export TMPDIR=/tmp
for c in "$@"; do
  rm -rf /tmp/prof_$c
  (cd /tmp && rocprofv3 --kernel-trace --stats -d /tmp/prof_$c -o $c -- python $GRAFT_REPO_ROOT/bench.py --config $c --serialize --no-graph --steps 3 --warmup 1 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/$tag/prof_$c.log 2>&1)
  db=$(find /tmp/prof_$c -name "*.db" | head -1)
  # 1 warm-up + 2 timed-gemm + 1 + 2 isolated + graph-less: warmup 1 + timed 3  => count steps from the log instead
  python probes/prof_summary.py $db 8 > gpurun_out/$tag/${c}_serialized_kernel_stats.txt
  head -30 gpurun_out/$tag/${c}_serialized_kernel_stats.txt
done
