mkdir -p gpurun_out/r3
python tools/dev/dbg_seed3.py > gpurun_out/r3/dbg_seed3.txt 2>&1; tail -30 gpurun_out/r3/dbg_seed3.txt
