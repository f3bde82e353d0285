set -x
mkdir -p gpurun_out/r3
( while sleep 45; do echo "tick $(date +%T)"; done ) &
TICK=$!
export TMPDIR=/tmp
python -m pytest tests -x -q -m gpu > gpurun_out/r3/run14_pytest_full.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r3/run14_pytest_full.log
tail -6 gpurun_out/r3/run14_pytest_full.log
MK_FUZZ_SEEDS=120 MK_SHARD_SEEDS=40 MK_STATE_SEEDS=120 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "randomised_cases or random_shardings or random_operation" > gpurun_out/r3/run14_soak.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r3/run14_soak.log
tail -4 gpurun_out/r3/run14_soak.log
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d gpurun_out/r3/pmc1 -o p -- python3 tools/build_rate.py 1280 20 > gpurun_out/r3/pmc1.log 2>&1
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_WR --kernel-trace --output-format csv -d gpurun_out/r3/pmc2 -o p -- python3 tools/build_rate.py 1280 20 > gpurun_out/r3/pmc2.log 2>&1
for d in pmc1 pmc2; do f=$(find gpurun_out/r3/$d -name '*counter_collection.csv' | head -1); python tools/pmc_summary.py $f build_ > gpurun_out/r3/run14_$d.txt 2>&1; rm -rf gpurun_out/r3/$d; done
cat gpurun_out/r3/run14_pmc1.txt gpurun_out/r3/run14_pmc2.txt
kill $TICK
