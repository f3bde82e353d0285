set -x
mkdir -p gpurun_out/r3
export TMPDIR=/tmp
python -m pytest tests -x -q -m gpu > gpurun_out/r3/run4_pytest_full.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r3/run4_pytest_full.log
tail -5 gpurun_out/r3/run4_pytest_full.log
for t in 1 2; do
  export MIEKKI_TUNE_BUILD=$t
  rocprofv3 --kernel-trace --stats -d gpurun_out/r3/bt$t -o d -- python3 tools/build_rate.py 6400 20 > gpurun_out/r3/bt$t.log 2>&1
  python tools/rocpd_stats.py gpurun_out/r3/bt$t/d_results.db > gpurun_out/r3/run4_build_t${t}_stats.csv 2>> gpurun_out/r3/bt$t.log
  rm -rf gpurun_out/r3/bt$t
  head -4 gpurun_out/r3/run4_build_t${t}_stats.csv
done
unset MIEKKI_TUNE_BUILD
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d gpurun_out/r3/pmc1 -o p -- python3 tools/build_rate.py 1280 20 > gpurun_out/r3/pmc1.log 2>&1
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_WR --kernel-trace --output-format csv -d gpurun_out/r3/pmc2 -o p -- python3 tools/build_rate.py 1280 20 > gpurun_out/r3/pmc2.log 2>&1
for d in pmc1 pmc2; do f=$(find gpurun_out/r3/$d -name '*counter_collection.csv' | head -1); python tools/pmc_summary.py $f build_ > gpurun_out/r3/run4_$d.txt 2>&1; rm -rf gpurun_out/r3/$d; done
cat gpurun_out/r3/run4_pmc1.txt gpurun_out/r3/run4_pmc2.txt
python bench.py > gpurun_out/r3/run4_bench.json 2> gpurun_out/r3/run4_bench.err; tail -3 gpurun_out/r3/run4_bench.err; cat gpurun_out/r3/run4_bench.json
