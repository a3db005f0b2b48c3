mkdir -p gpurun_out/r2h
X2_PARITY_DUMP=gpurun_out/r2h/parity timeout 900 python -m pytest tests -m gpu -q > gpurun_out/r2h/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r2h/pytest.log
tail -5 gpurun_out/r2h/pytest.log
bash probes/run_pmc.sh r02
