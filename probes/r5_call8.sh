#!/bin/bash
# round 5: the whole GPU suite (with the parity dump for profiles/r09_parity_worst.txt), smoke, then the driver's bench command
set -x
OUT=gpurun_out/r5c8
mkdir -p $OUT
export X2_PARITY_DUMP=$OUT/parity
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -x -q -m gpu --durations=15 > $OUT/pytest_gpu.log 2>&1; echo "rc=$?" >> $OUT/pytest_gpu.log
unset X2_PARITY_DUMP
grep -v "UserWarning\|Consider using\|return Variable\|^$\|amdgpu.ids" $OUT/pytest_gpu.log | tail -n 30 | cut -c1-300
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver_cmd.json 2> $OUT/bench_driver_cmd.err
cut -c1-700 $OUT/bench_driver_cmd.json
