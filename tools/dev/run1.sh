set -x
mkdir -p gpurun_out/r3
python -m pytest tests/test_gpu_cold_rows.py tests/test_gpu_cli.py tests/test_gpu_edges.py -x -q -m gpu > gpurun_out/r3/run1_pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r3/run1_pytest.log
tail -5 gpurun_out/r3/run1_pytest.log
./tools/ubench > gpurun_out/r3/ubench.txt 2>&1
tail -3 gpurun_out/r3/ubench.txt
export TMPDIR=/tmp
for t in 0 1 2 3; do
  export MIEKKI_TUNE_BUILD=$t
  rocprofv3 --kernel-trace --stats -d gpurun_out/r3/build_t$t -o d -- python3 tools/build_rate.py 6400 20 > gpurun_out/r3/build_t$t.log 2>&1
  python tools/rocpd_stats.py gpurun_out/r3/build_t$t/d_results.db > gpurun_out/r3/build_t${t}_stats.csv 2>> gpurun_out/r3/build_t$t.log
  rm -rf gpurun_out/r3/build_t$t
  tail -1 gpurun_out/r3/build_t$t.log; head -6 gpurun_out/r3/build_t${t}_stats.csv
done
