set -x
mkdir -p gpurun_out/r3
( while sleep 45; do echo "tick $(date +%T)"; done ) &
TICK=$!
export TMPDIR=/tmp
python tools/build_rate.py 6400 20 > gpurun_out/r3/run12_build_rate.txt 2>&1; cat gpurun_out/r3/run12_build_rate.txt
python -m pytest tests/test_gpu_packed.py tests/test_gpu_parity.py tests/test_gpu_edges.py -x -q -m gpu > gpurun_out/r3/run12_pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r3/run12_pytest.log
tail -8 gpurun_out/r3/run12_pytest.log
rocprofv3 --kernel-trace --stats -d gpurun_out/r3/b12 -o d -- python3 tools/build_rate.py 6400 20 > gpurun_out/r3/b12.log 2>&1
python tools/rocpd_stats.py gpurun_out/r3/b12/d_results.db > gpurun_out/r3/run12_build_stats.csv 2>> gpurun_out/r3/b12.log
rm -rf gpurun_out/r3/b12
head -7 gpurun_out/r3/run12_build_stats.csv
for t in 1 2; do MIEKKI_TUNE_BUILD=$t python tools/build_rate.py 6400 20 2>&1 | tail -1; done
kill $TICK
