set -x
mkdir -p gpurun_out/r3
( while sleep 45; do echo "tick $(date +%T)"; done ) &
TICK=$!
export TMPDIR=/tmp
for n in 1 2 3 4; do
  MIEKKI_COPY_STREAMS=$n timeout -k 10 120 python tools/host_fed_rate.py 24 > gpurun_out/r3/run20_hostfed_$n.txt 2>&1; tail -2 gpurun_out/r3/run20_hostfed_$n.txt
done
timeout -k 10 240 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum --kernel-trace --output-format csv -d gpurun_out/r3/pmc3 -o p -- python3 tools/build_rate.py 1280 20 > gpurun_out/r3/pmc3.log 2>&1
f=$(find gpurun_out/r3/pmc3 -name '*counter_collection.csv' | head -1); python tools/pmc_summary.py $f build_ > gpurun_out/r3/run20_pmc3.txt 2>&1; rm -rf gpurun_out/r3/pmc3
cat gpurun_out/r3/run20_pmc3.txt
timeout -k 10 500 python bench.py > gpurun_out/r3/run20_bench.json 2> gpurun_out/r3/run20_bench.err; echo "bench rc=$?"
cat gpurun_out/r3/run20_bench.json
kill $TICK
