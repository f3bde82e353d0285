mkdir -p gpurun_out/r3
MK_DBG_NOFORGET=1 timeout -k 10 300 python -m pytest tests/test_gpu_packed.py -x -q -m gpu -k settled > gpurun_out/r3/run35_pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r3/run35_pytest.log
tail -12 gpurun_out/r3/run35_pytest.log
