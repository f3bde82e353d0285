set -x
mkdir -p gpurun_out/r3
( while sleep 45; do echo "tick $(date +%T)"; done ) &
TICK=$!
export TMPDIR=/tmp
MK_FUZZ_SEEDS=200 MK_SHARD_SEEDS=150 MK_STATE_SEEDS=150 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "randomised_cases or random_shardings or random_operation" > gpurun_out/r3/run11_soak.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r3/run11_soak.log
tail -6 gpurun_out/r3/run11_soak.log
python tools/cli_e2e.py 256 20000 16 > gpurun_out/r3/run11_cli_e2e.txt 2>&1; cat gpurun_out/r3/run11_cli_e2e.txt
python tools/latency.py > gpurun_out/r3/run11_latency.txt 2>&1; tail -3 gpurun_out/r3/run11_latency.txt
kill $TICK
