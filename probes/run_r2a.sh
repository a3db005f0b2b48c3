mkdir -p gpurun_out/r2a
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/r2a/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r2a/pytest.log
tail -5 gpurun_out/r2a/pytest.log
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r2a/bench_base.log 2>&1; tail -c 1500 gpurun_out/r2a/bench_base.log
timeout 400 python bench.py --config large --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/r2a/bench_large.log 2>&1; tail -c 1500 gpurun_out/r2a/bench_large.log
timeout 300 python bench.py --config video --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/r2a/bench_video.log 2>&1; tail -c 1500 gpurun_out/r2a/bench_video.log
