mkdir -p gpurun_out/r3
timeout -k 10 200 python tools/dev/dbg_bloom.py > gpurun_out/r3/run27_dbg.txt 2>&1
cat gpurun_out/r3/run27_dbg.txt | tail -20
