set -x
mkdir -p gpurun_out/r3
( while sleep 45; do echo "tick $(date +%T)"; done ) &
TICK=$!
export TMPDIR=/tmp
python tools/build_rate.py 6400 20 > gpurun_out/r3/run10_build_rate.txt 2>&1; cat gpurun_out/r3/run10_build_rate.txt
python -m pytest tests -x -q -m gpu > gpurun_out/r3/run10_pytest_full.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r3/run10_pytest_full.log
tail -8 gpurun_out/r3/run10_pytest_full.log
rocprofv3 --kernel-trace --stats -d gpurun_out/r3/b10 -o d -- python3 tools/build_rate.py 6400 20 > gpurun_out/r3/b10.log 2>&1
python tools/rocpd_stats.py gpurun_out/r3/b10/d_results.db > gpurun_out/r3/run10_build_stats.csv 2>> gpurun_out/r3/b10.log
rm -rf gpurun_out/r3/b10
head -9 gpurun_out/r3/run10_build_stats.csv
kill $TICK
