set -x
mkdir -p gpurun_out/r3
( while sleep 45; do echo "tick $(date +%T)"; done ) &
TICK=$!
export TMPDIR=/tmp
timeout -k 10 100 python tools/build_rate.py 6400 20 > gpurun_out/r3/run30_build_rate.txt 2>&1; tail -1 gpurun_out/r3/run30_build_rate.txt
timeout -k 10 100 python tools/build_rate.py 25600 20 > gpurun_out/r3/run30_build_rate_25600.txt 2>&1; tail -1 gpurun_out/r3/run30_build_rate_25600.txt
timeout -k 10 200 rocprofv3 --kernel-trace --stats -d gpurun_out/r3/b30 -o d -- python3 tools/build_rate.py 12800 20 > gpurun_out/r3/b30.log 2>&1
python tools/rocpd_stats.py gpurun_out/r3/b30/d_results.db > gpurun_out/r3/run30_build_stats.csv 2>&1
python tools/rocpd_timeline.py gpurun_out/r3/b30/d_results.db 2700 60 > gpurun_out/r3/run30_timeline.txt 2>&1
rm -rf gpurun_out/r3/b30
timeout -k 10 240 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d gpurun_out/r3/pmc1 -o p -- python3 tools/build_rate.py 6400 40 > gpurun_out/r3/pmc1.log 2>&1
timeout -k 10 240 rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_WR --kernel-trace --output-format csv -d gpurun_out/r3/pmc2 -o p -- python3 tools/build_rate.py 6400 40 > gpurun_out/r3/pmc2.log 2>&1
timeout -k 10 240 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum --kernel-trace --output-format csv -d gpurun_out/r3/pmc3 -o p -- python3 tools/build_rate.py 6400 40 > gpurun_out/r3/pmc3.log 2>&1
for d in pmc1 pmc2 pmc3; do f=$(find gpurun_out/r3/$d -name '*counter_collection.csv' | head -1); python tools/pmc_summary.py $f build_ 60 > gpurun_out/r3/run30_$d.txt 2>&1; rm -rf gpurun_out/r3/$d; done
cat gpurun_out/r3/run30_pmc1.txt gpurun_out/r3/run30_pmc2.txt gpurun_out/r3/run30_pmc3.txt
timeout -k 10 400 python bench.py > gpurun_out/r3/run30_bench.json 2> gpurun_out/r3/run30_bench.err; echo "bench rc=$?"
cat gpurun_out/r3/run30_bench.json
kill $TICK
