set -x
mkdir -p gpurun_out/r3
( while sleep 45; do echo "tick $(date +%T)"; done ) &
TICK=$!
export TMPDIR=/tmp
python -m pytest tests/test_gpu_packed.py tests/test_gpu_edges.py -x -q -m gpu > gpurun_out/r3/run16_pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r3/run16_pytest.log
tail -3 gpurun_out/r3/run16_pytest.log
for t in 0 2 1 3; do
  MIEKKI_TUNE_BUILD=$t python tools/build_rate.py 6400 20 > gpurun_out/r3/run16_rate_t$t.txt 2>&1; tail -1 gpurun_out/r3/run16_rate_t$t.txt
done
rocprofv3 --kernel-trace --stats -d gpurun_out/r3/b16 -o d -- python3 tools/build_rate.py 6400 20 > gpurun_out/r3/b16.log 2>&1
python tools/rocpd_stats.py gpurun_out/r3/b16/d_results.db > gpurun_out/r3/run16_build_stats.csv 2>&1; rm -rf gpurun_out/r3/b16
cut -c1-50,150-400 gpurun_out/r3/run16_build_stats.csv | head -6
kill $TICK
