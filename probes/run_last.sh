mkdir -p gpurun_out/last
timeout 12 python probes/bench_epi_f4.py > gpurun_out/last/epi_f4.log 2>&1
tail -10 gpurun_out/last/epi_f4.log
timeout 14 python -m pytest tests/test_kernels_gpu.py -m gpu -q -x -k "row_contiguous" > gpurun_out/last/pytest_f4.log 2>&1
tail -3 gpurun_out/last/pytest_f4.log
