set -x
mkdir -p gpurun_out/r3
( while sleep 45; do echo "tick $(date +%T)"; done ) &
TICK=$!
export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests -x -q -m gpu > gpurun_out/r3/run28_pytest_full.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r3/run28_pytest_full.log
tail -4 gpurun_out/r3/run28_pytest_full.log
MK_FUZZ_SEEDS=100 MK_SHARD_SEEDS=40 MK_STATE_SEEDS=100 timeout -k 10 480 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "randomised_cases or random_shardings or random_operation" > gpurun_out/r3/run28_soak.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r3/run28_soak.log
tail -4 gpurun_out/r3/run28_soak.log
kill $TICK
