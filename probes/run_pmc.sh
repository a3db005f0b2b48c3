# usage: bash probes/run_pmc.sh <tag>  -> gpurun_out/<tag>/{<tag>_pmc_FETCH_SIZE.txt, <tag>_pmc_WRITE_SIZE.txt, pmc_traffic.json}
tag=$1
mkdir -p gpurun_out/$tag
export TMPDIR=/tmp
for ctr in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$ctr
  (cd /tmp && timeout 900 rocprofv3 --pmc $ctr -d /tmp/pmc_$ctr -o p -- python $GRAFT_REPO_ROOT/bench.py --serialize --no-graph --steps 2 --warmup 1 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/$tag/pmc_$ctr.log 2>&1)
  db=$(find /tmp/pmc_$ctr -name "*.db" | head -1)
  python probes/pmc_summary.py $db $ctr > gpurun_out/$tag/${tag}_pmc_$ctr.txt
  head -8 gpurun_out/$tag/${tag}_pmc_$ctr.txt
done
python probes/pmc_traffic.py gpurun_out/$tag/${tag}_pmc_FETCH_SIZE.txt gpurun_out/$tag/${tag}_pmc_WRITE_SIZE.txt $tag > gpurun_out/$tag/pmc_traffic.json
cat gpurun_out/$tag/pmc_traffic.json
