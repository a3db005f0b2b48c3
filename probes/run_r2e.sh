mkdir -p gpurun_out/r2e
X2_GRAPH_TRACE=1 timeout 300 python -X faulthandler bench.py --tiny --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r2e/tiny.log 2>&1; grep -v "Warning\|warn" gpurun_out/r2e/tiny.log | tail -c 1500 | cut -c1-600
for c in base video large; do
timeout 400 python -X faulthandler bench.py --config $c --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/r2e/bench_$c.log 2>&1; grep -v "Warning\|warn" gpurun_out/r2e/bench_$c.log | tail -c 2800 | cut -c1-900; echo
done
timeout 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --with-optimizer > gpurun_out/r2e/bench_base_opt.log 2>&1; tail -c 600 gpurun_out/r2e/bench_base_opt.log
