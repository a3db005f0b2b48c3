mkdir -p gpurun_out/r2b
timeout 600 python -m pytest tests/test_optim_gpu.py tests/test_train_mode_gpu.py tests/test_kernels_gpu.py -x -q > gpurun_out/r2b/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r2b/pytest.log
tail -4 gpurun_out/r2b/pytest.log
for c in base video large; do
timeout 400 python bench.py --config $c --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/r2b/bench_$c.log 2>&1; tail -c 2500 gpurun_out/r2b/bench_$c.log; echo
done
timeout 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-graph > gpurun_out/r2b/bench_base_nograph.log 2>&1; tail -c 600 gpurun_out/r2b/bench_base_nograph.log
X2_DDP_TEST=1 timeout 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --with-optimizer > gpurun_out/r2b/bench_base_opt.log 2>&1; tail -c 600 gpurun_out/r2b/bench_base_opt.log
