set -x
mkdir -p gpurun_out/r3
( while sleep 45; do echo "tick $(date +%T)"; done ) &
TICK=$!
export TMPDIR=/tmp
python tools/build_rate.py 6400 20 > gpurun_out/r3/run7_build_rate.txt 2>&1; cat gpurun_out/r3/run7_build_rate.txt
python -m pytest tests -x -q -m gpu > gpurun_out/r3/run7_pytest_full.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r3/run7_pytest_full.log
tail -8 gpurun_out/r3/run7_pytest_full.log
rocprofv3 --kernel-trace --stats -d gpurun_out/r3/b7 -o d -- python3 tools/build_rate.py 6400 20 > gpurun_out/r3/b7.log 2>&1
python tools/rocpd_stats.py gpurun_out/r3/b7/d_results.db > gpurun_out/r3/run7_build_stats.csv 2>> gpurun_out/r3/b7.log
rm -rf gpurun_out/r3/b7
head -6 gpurun_out/r3/run7_build_stats.csv
python bench.py --no-cpu-baseline > gpurun_out/r3/run7_bench.json 2> gpurun_out/r3/run7_bench.err; python -c "
import json; d=json.load(open('gpurun_out/r3/run7_bench.json')); print(d['value'], d['sketch'])"
python tools/ingest_bench.py 4096 32 64 > gpurun_out/r3/run7_ingest_32.txt 2>&1; cat gpurun_out/r3/run7_ingest_32.txt
MIEKKI_INGEST=chars python tools/ingest_bench.py 2048 32 64 > gpurun_out/r3/run7_ingest_32_chars.txt 2>&1; tail -4 gpurun_out/r3/run7_ingest_32_chars.txt
python tools/bench_dense.py 100000 32 > gpurun_out/r3/run7_dense_hbm.json 2>&1; tail -2 gpurun_out/r3/run7_dense_hbm.json
MIEKKI_HBM_MATRIX_MIB=75000 python tools/bench_dense.py 100000 32 > gpurun_out/r3/run7_dense_cold.json 2>&1; tail -2 gpurun_out/r3/run7_dense_cold.json
MIEKKI_HBM_MATRIX_MIB=75000 python bench.py --no-cpu-baseline --steps 1 --warmup 1 > gpurun_out/r3/run7_bench_cold.json 2> gpurun_out/r3/run7_bench_cold.err; python -c "
import json; d=json.load(open('gpurun_out/r3/run7_bench_cold.json')); print(d['value'], d['ms_per_step'], d['check'])"
kill $TICK
