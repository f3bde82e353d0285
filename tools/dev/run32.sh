set -x
mkdir -p gpurun_out/r3
( while sleep 45; do echo "tick $(date +%T)"; done ) &
TICK=$!
export TMPDIR=/tmp
timeout -k 10 400 python -m pytest tests/test_gpu_packed.py tests/test_gpu_edges.py tests/test_gpu_cold_rows.py -x -q -m gpu > gpurun_out/r3/run32_pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r3/run32_pytest.log
tail -3 gpurun_out/r3/run32_pytest.log
for rep in 1 2; do timeout -k 10 100 python tools/build_rate.py 12800 20 > gpurun_out/r3/run32_rate_$rep.txt 2>&1; tail -1 gpurun_out/r3/run32_rate_$rep.txt; done
timeout -k 10 120 python tools/host_fed_rate.py 24 > gpurun_out/r3/run32_hostfed.txt 2>&1; tail -2 gpurun_out/r3/run32_hostfed.txt
timeout -k 10 200 rocprofv3 --kernel-trace --stats -d gpurun_out/r3/b32 -o d -- python3 tools/build_rate.py 12800 20 > gpurun_out/r3/b32.log 2>&1
python tools/rocpd_timeline.py gpurun_out/r3/b32/d_results.db 2700 45 > gpurun_out/r3/run32_timeline.txt 2>&1
rm -rf gpurun_out/r3/b32
cat gpurun_out/r3/run32_timeline.txt
MK_STATE_SEEDS=60 timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "random_operation" > gpurun_out/r3/run32_soak.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r3/run32_soak.log
tail -3 gpurun_out/r3/run32_soak.log
kill $TICK
