mkdir -p gpurun_out/last
timeout 16 python probes/bench_epi_swap.py > gpurun_out/last/epi_swap.log 2>&1
tail -12 gpurun_out/last/epi_swap.log
timeout 28 python -m pytest tests/test_kernels_gpu.py -m gpu -q -x -k "lane_swap" > gpurun_out/last/pytest_swap.log 2>&1
tail -4 gpurun_out/last/pytest_swap.log
