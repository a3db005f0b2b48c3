for v in 0 256 1024 512 2048 768 3072; do echo "== X2_ATTN_VARIANT=$v"; X2_ATTN_VARIANT=$v python probes/bench_attn.py 2>&1 | grep "vision large"; done
timeout 300 python -m pytest tests/test_kernels_gpu.py -q -k attention 2>&1 | tail -2
