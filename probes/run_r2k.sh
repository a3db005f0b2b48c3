mkdir -p gpurun_out/r2k
timeout 900 python -m pytest tests -m gpu -q > gpurun_out/r2k/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r2k/pytest.log
grep -E "passed|failed|rc=|whole-model train" gpurun_out/r2k/pytest.log | tail -5
X2_BENCH_BACKEND=gloo timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --tiny --steps 3 --warmup 1 > gpurun_out/r2k/bench_2rank_gloo.log 2>&1; echo "2-rank rc=$?"; grep '^{' gpurun_out/r2k/bench_2rank_gloo.log | cut -c1-400
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
