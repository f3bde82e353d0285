set -x
mkdir -p gpurun_out/r3
( while sleep 45; do echo "tick $(date +%T)"; done ) &
TICK=$!
export TMPDIR=/tmp
python -m pytest tests -x -q -m gpu > gpurun_out/r3/run19_pytest_full.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r3/run19_pytest_full.log
tail -6 gpurun_out/r3/run19_pytest_full.log
python tools/build_rate.py 6400 20 > gpurun_out/r3/run19_build_rate.txt 2>&1; cat gpurun_out/r3/run19_build_rate.txt
rocprofv3 --kernel-trace --stats -d gpurun_out/r3/b19 -o d -- python3 tools/build_rate.py 6400 20 > gpurun_out/r3/b19.log 2>&1
python tools/rocpd_stats.py gpurun_out/r3/b19/d_results.db > gpurun_out/r3/run19_build_stats.csv 2>&1
python tools/rocpd_timeline.py gpurun_out/r3/b19/d_results.db 900 60 > gpurun_out/r3/run19_timeline.txt 2>&1
rm -rf gpurun_out/r3/b19
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d gpurun_out/r3/pmc1 -o p -- python3 tools/build_rate.py 1280 20 > gpurun_out/r3/pmc1.log 2>&1
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_WR --kernel-trace --output-format csv -d gpurun_out/r3/pmc2 -o p -- python3 tools/build_rate.py 1280 20 > gpurun_out/r3/pmc2.log 2>&1
rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum SQ_INSTS_VMEM_RD --kernel-trace --output-format csv -d gpurun_out/r3/pmc3 -o p -- python3 tools/build_rate.py 1280 20 > gpurun_out/r3/pmc3.log 2>&1
for d in pmc1 pmc2 pmc3; do f=$(find gpurun_out/r3/$d -name '*counter_collection.csv' | head -1); python tools/pmc_summary.py $f build_ > gpurun_out/r3/run19_$d.txt 2>&1; rm -rf gpurun_out/r3/$d; done
cat gpurun_out/r3/run19_pmc1.txt gpurun_out/r3/run19_pmc2.txt gpurun_out/r3/run19_pmc3.txt
kill $TICK
