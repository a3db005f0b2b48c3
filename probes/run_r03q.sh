#!/bin/bash
# round 3 final evidence: full GPU suite, serialized kernel traces (base / large / video), PMC passes (HBM traffic, MFMA busy)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03q
timeout 1800 python -m pytest tests -q -m gpu --maxfail=8 > gpurun_out/r03q/pytest_all.log 2>&1; echo "rc=$?" >> gpurun_out/r03q/pytest_all.log
tail -n 4 gpurun_out/r03q/pytest_all.log | cut -c1-300
bash probes/run_prof.sh r03q base large video > gpurun_out/r03q/prof.log 2>&1
head -6 gpurun_out/r03q/base_serialized_kernel_stats.txt
bash probes/run_pmc.sh r03q > gpurun_out/r03q/pmc.log 2>&1; tail -n 3 gpurun_out/r03q/pmc.log
(cd /tmp && rocprofv3 -L 2>/dev/null | grep -i -E "mfma|MfmaUtil|VALUBusy|SQ_BUSY_CYCLES|SQ_WAVE_CYCLES|GRBM_GUI_ACTIVE" | head -40) > gpurun_out/r03q/counters_avail.txt 2>&1
head -30 gpurun_out/r03q/counters_avail.txt
for ctr in SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES; do
  rm -rf /tmp/pmc_$ctr
  (cd /tmp && timeout 600 rocprofv3 --pmc $ctr -d /tmp/pmc_$ctr -o p -- python $GRAFT_REPO_ROOT/bench.py --serialize --no-graph --steps 2 --warmup 1 --no-cpu-baseline --no-other-configs > $GRAFT_REPO_ROOT/gpurun_out/r03q/pmc_$ctr.log 2>&1)
  db=$(find /tmp/pmc_$ctr -name "*.db" | head -1)
  [ -n "$db" ] && python probes/pmc_summary.py $db $ctr > gpurun_out/r03q/r03q_pmc_$ctr.txt && head -8 gpurun_out/r03q/r03q_pmc_$ctr.txt | cut -c1-140
done
