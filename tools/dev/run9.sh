set -x
mkdir -p gpurun_out/r3
( while sleep 45; do echo "tick $(date +%T)"; done ) &
TICK=$!
export TMPDIR=/tmp
( echo "nproc $(nproc)"; echo "cpu.max: $(cat /sys/fs/cgroup/cpu.max 2>&1)"; echo "v1 quota: $(cat /sys/fs/cgroup/cpu/cpu.cfs_quota_us 2>&1) period $(cat /sys/fs/cgroup/cpu/cpu.cfs_period_us 2>&1)"; cat /proc/self/cgroup; grep -i cpus_allowed_list /proc/self/status; python -c "import sys; sys.path.insert(0,'.'); sys.argv=['x']; import bench; print('usable', bench.usable_cpus())" ) > gpurun_out/r3/run9_cpus.txt 2>&1; cat gpurun_out/r3/run9_cpus.txt
python tools/ingest_bench.py 16384 16 64 > gpurun_out/r3/run9_ingest_16.txt 2>&1; cat gpurun_out/r3/run9_ingest_16.txt
python tools/ingest_bench.py 16384 32 64 > gpurun_out/r3/run9_ingest_32.txt 2>&1; tail -4 gpurun_out/r3/run9_ingest_32.txt
python tools/column_entropy.py 4096 20 4096 > gpurun_out/r3/run9_column_entropy.txt 2>&1; cat gpurun_out/r3/run9_column_entropy.txt
rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_DRAM_sum TCC_EA0_RDREQ_32B_sum --kernel-trace --output-format csv -d gpurun_out/r3/pmc3 -o p -- python3 bench.py --genomes 12500 --queries 21000 --steps 1 --warmup 0 --no-cpu-baseline > gpurun_out/r3/pmc3.log 2>&1
f=$(find gpurun_out/r3/pmc3 -name '*counter_collection.csv' | head -1); python tools/pmc_summary.py $f scan_slab > gpurun_out/r3/run9_pmc_dram.txt 2>&1; rm -rf gpurun_out/r3/pmc3; cat gpurun_out/r3/run9_pmc_dram.txt
python bench.py > gpurun_out/r3/run9_bench.json 2> gpurun_out/r3/run9_bench.err; tail -3 gpurun_out/r3/run9_bench.err; python -c "
import json; d=json.load(open('gpurun_out/r3/run9_bench.json')); print(d['value'], d['roofline']['frac']); print(d['sketch']); print(d['cpu_baseline'])"
kill $TICK
