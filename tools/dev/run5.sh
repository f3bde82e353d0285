set -x
mkdir -p gpurun_out/r3
( while sleep 45; do echo "tick $(date +%T)"; done ) &
TICK=$!
export TMPDIR=/tmp
python -m pytest tests/test_gpu_cold_rows.py tests/test_gpu_packed.py -x -q -m gpu > gpurun_out/r3/run5_pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r3/run5_pytest.log
tail -5 gpurun_out/r3/run5_pytest.log
python tools/build_rate.py 6400 20 > gpurun_out/r3/run5_build_rate.txt 2>&1; cat gpurun_out/r3/run5_build_rate.txt
rocprofv3 --kernel-trace --stats -d gpurun_out/r3/b5 -o d -- python3 tools/build_rate.py 6400 20 > gpurun_out/r3/b5.log 2>&1
python tools/rocpd_stats.py gpurun_out/r3/b5/d_results.db > gpurun_out/r3/run5_build_stats.csv 2>> gpurun_out/r3/b5.log
rm -rf gpurun_out/r3/b5
head -8 gpurun_out/r3/run5_build_stats.csv
nproc; free -g | head -2; lscpu | grep -E "Model name|Socket|NUMA node\(s\)|Thread"
H=oracle/_ref/ref_harness
( time timeout 400 $H scanbench 20 20000 1 64 ) > gpurun_out/r3/run5_scan_64.txt 2>&1; tail -4 gpurun_out/r3/run5_scan_64.txt
( export MALLOC_MMAP_THRESHOLD_=4294967296 MALLOC_TRIM_THRESHOLD_=17179869184 MALLOC_TOP_PAD_=268435456; time timeout 400 $H scanbench 20 20000 1 64 ) > gpurun_out/r3/run5_scan_64_malloc.txt 2>&1; tail -4 gpurun_out/r3/run5_scan_64_malloc.txt
( export MALLOC_MMAP_THRESHOLD_=4294967296 MALLOC_TRIM_THRESHOLD_=17179869184 MALLOC_TOP_PAD_=268435456; time timeout 500 $H scanbench 20 20000 1 256 ) > gpurun_out/r3/run5_scan_256_malloc.txt 2>&1; tail -4 gpurun_out/r3/run5_scan_256_malloc.txt
( time timeout 300 $H sketchbench 20 128 128 ) > gpurun_out/r3/run5_sketch_128.txt 2>&1; tail -4 gpurun_out/r3/run5_sketch_128.txt
kill $TICK
