set -x
mkdir -p gpurun_out/r3
( while sleep 45; do echo "tick $(date +%T)"; done ) &
TICK=$!
export TMPDIR=/tmp
python -m pytest tests/test_gpu_cold_rows.py tests/test_gpu_cli.py -x -q -m gpu > gpurun_out/r3/run8_pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r3/run8_pytest.log
tail -4 gpurun_out/r3/run8_pytest.log
python tools/ingest_bench.py 16384 32 64 > gpurun_out/r3/run8_ingest_32.txt 2>&1; cat gpurun_out/r3/run8_ingest_32.txt
python tools/ingest_bench.py 16384 64 64 > gpurun_out/r3/run8_ingest_64.txt 2>&1; tail -4 gpurun_out/r3/run8_ingest_64.txt
MIEKKI_INGEST=chars python tools/ingest_bench.py 8192 32 64 > gpurun_out/r3/run8_ingest_32_chars.txt 2>&1; tail -4 gpurun_out/r3/run8_ingest_32_chars.txt
MIEKKI_HBM_MATRIX_MIB=75000 python tools/bench_dense.py 100000 32 > gpurun_out/r3/run8_dense_cold.json 2>&1; tail -1 gpurun_out/r3/run8_dense_cold.json
python bench.py --genomes 10000 --fp-bits 16 --no-cpu-baseline > gpurun_out/r3/run8_c4_default.json 2>/dev/null; python -c "
import json; d=json.load(open('gpurun_out/r3/run8_c4_default.json')); print('c4 default', d['value'], d['roofline']['frac'], d['roofline']['launches'], d['select'])"
MIEKKI_SLAB_MAX_QUERIES=25000 python bench.py --genomes 10000 --fp-bits 16 --no-cpu-baseline > gpurun_out/r3/run8_c4_q25k.json 2>/dev/null; python -c "
import json; d=json.load(open('gpurun_out/r3/run8_c4_q25k.json')); print('c4 25k', d['value'], d['roofline']['frac'], d['roofline']['launches'], d['select'])"
MIEKKI_SLAB_MAX_QUERIES=12500 python bench.py --genomes 10000 --fp-bits 16 --no-cpu-baseline > gpurun_out/r3/run8_c4_q12k.json 2>/dev/null; python -c "
import json; d=json.load(open('gpurun_out/r3/run8_c4_q12k.json')); print('c4 12.5k', d['value'], d['roofline']['frac'], d['roofline']['launches'], d['select'])"
rocprofv3 -L > gpurun_out/r3/run8_counters_all.txt 2>&1; grep -i -E "mall|umc|hbm|dram|MC_|EA_RD|EA0" gpurun_out/r3/run8_counters_all.txt | cut -c1-200 | head -40 > gpurun_out/r3/run8_counters_mem.txt; wc -l gpurun_out/r3/run8_counters_all.txt; head -30 gpurun_out/r3/run8_counters_mem.txt
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d gpurun_out/r3/pmc1 -o p -- python3 tools/build_rate.py 1280 20 > gpurun_out/r3/pmc1.log 2>&1
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_WR --kernel-trace --output-format csv -d gpurun_out/r3/pmc2 -o p -- python3 tools/build_rate.py 1280 20 > gpurun_out/r3/pmc2.log 2>&1
for d in pmc1 pmc2; do f=$(find gpurun_out/r3/$d -name '*counter_collection.csv' | head -1); python tools/pmc_summary.py $f build_ > gpurun_out/r3/run8_$d.txt 2>&1; rm -rf gpurun_out/r3/$d; done
cat gpurun_out/r3/run8_pmc1.txt gpurun_out/r3/run8_pmc2.txt
kill $TICK
