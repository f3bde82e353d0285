set -x
mkdir -p gpurun_out/r3
( while sleep 45; do echo "tick $(date +%T)"; done ) &
TICK=$!
export TMPDIR=/tmp
timeout -k 10 300 python bench.py --genomes 10000 --fp-bits 16 --no-cpu-baseline > gpurun_out/r3/run33_bench_c4.json 2> gpurun_out/r3/run33_bench_c4.err; echo "rc=$?"
tail -c 1500 gpurun_out/r3/run33_bench_c4.json
timeout -k 10 300 python bench.py --genomes 1000 --queries 10000 --h 17 --no-cpu-baseline > gpurun_out/r3/run33_bench_c2.json 2> gpurun_out/r3/run33_bench_c2.err; echo "rc=$?"
tail -c 1500 gpurun_out/r3/run33_bench_c2.json
timeout -k 10 600 python -m pytest tests -x -q -m gpu > gpurun_out/r3/run33_pytest_full.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r3/run33_pytest_full.log
tail -4 gpurun_out/r3/run33_pytest_full.log
kill $TICK
