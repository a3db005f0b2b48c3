mkdir -p gpurun_out/r2c
for c in base video large; do
timeout 400 python bench.py --config $c --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/r2c/bench_$c.log 2>&1; grep -v Warning gpurun_out/r2c/bench_$c.log | tail -c 1800 ; echo
done
timeout 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --with-optimizer > gpurun_out/r2c/bench_base_opt.log 2>&1; tail -c 600 gpurun_out/r2c/bench_base_opt.log
