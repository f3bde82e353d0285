set -x
mkdir -p gpurun_out/r3
( while sleep 45; do echo "tick $(date +%T)"; done ) &
TICK=$!
export TMPDIR=/tmp
python tools/build_rate.py 6400 20 > gpurun_out/r3/run13_build_rate.txt 2>&1; cat gpurun_out/r3/run13_build_rate.txt
for t in 1 2; do MIEKKI_TUNE_BUILD=$t python tools/build_rate.py 6400 20 2>&1 | tail -1; done
rocprofv3 --kernel-trace --stats -d gpurun_out/r3/b13 -o d -- python3 tools/build_rate.py 6400 20 > gpurun_out/r3/b13.log 2>&1
python tools/rocpd_stats.py gpurun_out/r3/b13/d_results.db > gpurun_out/r3/run13_build_stats.csv 2>> gpurun_out/r3/b13.log
rm -rf gpurun_out/r3/b13
head -5 gpurun_out/r3/run13_build_stats.csv
python -m pytest tests/test_gpu_packed.py tests/test_gpu_edges.py -x -q -m gpu > gpurun_out/r3/run13_pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r3/run13_pytest.log
tail -4 gpurun_out/r3/run13_pytest.log
kill $TICK
