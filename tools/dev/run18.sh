set -x
mkdir -p gpurun_out/r3
( while sleep 45; do echo "tick $(date +%T)"; done ) &
TICK=$!
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d gpurun_out/r3/b18 -o d -- python3 tools/build_rate.py 6400 20 > gpurun_out/r3/b18.log 2>&1
python tools/rocpd_stats.py gpurun_out/r3/b18/d_results.db > gpurun_out/r3/run18_build_stats.csv 2>&1
python tools/rocpd_timeline.py gpurun_out/r3/b18/d_results.db 900 80 > gpurun_out/r3/run18_timeline.txt 2>&1
rm -rf gpurun_out/r3/b18
cat gpurun_out/r3/run18_timeline.txt
python -m pytest tests/test_gpu_packed.py tests/test_gpu_edges.py tests/test_gpu_parity.py -x -q -m gpu > gpurun_out/r3/run18_pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r3/run18_pytest.log
tail -3 gpurun_out/r3/run18_pytest.log
kill $TICK
