mkdir -p gpurun_out/r3
timeout -k 10 500 python -m pytest tests/test_gpu_packed.py -x -q -m gpu > gpurun_out/r3/run34_pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r3/run34_pytest.log
tail -30 gpurun_out/r3/run34_pytest.log
