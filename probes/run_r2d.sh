mkdir -p gpurun_out/r2d
X2_GRAPH_TRACE=1 timeout 300 python -X faulthandler bench.py --tiny --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r2d/tiny.log 2>&1; grep -v "Warning\|warn" gpurun_out/r2d/tiny.log | tail -c 3000
echo ==== serialized
X2_GRAPH_TRACE=1 timeout 300 python -X faulthandler bench.py --tiny --serialize --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r2d/tiny_ser.log 2>&1; grep -v "Warning\|warn" gpurun_out/r2d/tiny_ser.log | tail -c 3000
