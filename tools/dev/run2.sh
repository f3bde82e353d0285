set -x
mkdir -p gpurun_out/r3
python -m pytest tests/test_gpu_packed.py tests/test_gpu_parity.py -x -q -m gpu -k "packed or index_build or scores_bit or parameter_sweep or ragged or dump_and_load" > gpurun_out/r3/run2_pytest_a.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r3/run2_pytest_a.log
tail -15 gpurun_out/r3/run2_pytest_a.log
python tools/build_rate.py 6400 20 > gpurun_out/r3/run2_build_rate.txt 2>&1; cat gpurun_out/r3/run2_build_rate.txt
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d gpurun_out/r3/b2 -o d -- python3 tools/build_rate.py 6400 20 > gpurun_out/r3/b2.log 2>&1
python tools/rocpd_stats.py gpurun_out/r3/b2/d_results.db > gpurun_out/r3/run2_build_stats.csv 2>> gpurun_out/r3/b2.log
rm -rf gpurun_out/r3/b2
cat gpurun_out/r3/run2_build_stats.csv
python -m pytest tests -x -q -m gpu > gpurun_out/r3/run2_pytest_full.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r3/run2_pytest_full.log
tail -15 gpurun_out/r3/run2_pytest_full.log
