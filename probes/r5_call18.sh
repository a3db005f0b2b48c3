#!/bin/bash
# closing set on the final tree: kernel traces (base / large / video), PMC bytes, the driver's bench command
set -x
OUT=gpurun_out/r5c18
mkdir -p $OUT
bash probes/run_prof.sh r10a base large video > $OUT/run_prof.log 2>&1
bash probes/run_pmc.sh r10a > $OUT/run_pmc.log 2>&1
tail -12 $OUT/run_pmc.log
python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver_cmd.json 2> $OUT/bench_driver_cmd.err
cut -c1-500 $OUT/bench_driver_cmd.json
