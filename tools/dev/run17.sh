set -x
mkdir -p gpurun_out/r3
( while sleep 45; do echo "tick $(date +%T)"; done ) &
TICK=$!
export TMPDIR=/tmp
for t in 0 4 0 4; do
  MIEKKI_TUNE_BUILD=$t python tools/build_rate.py 6400 20 > gpurun_out/r3/run17_rate_t$t.txt 2>&1; tail -1 gpurun_out/r3/run17_rate_t$t.txt
done
MIEKKI_TUNE_BUILD=4 rocprofv3 --kernel-trace --stats -d gpurun_out/r3/b17 -o d -- python3 tools/build_rate.py 6400 20 > gpurun_out/r3/b17.log 2>&1
python tools/rocpd_stats.py gpurun_out/r3/b17/d_results.db > gpurun_out/r3/run17_build_stats.csv 2>&1; rm -rf gpurun_out/r3/b17
cut -c1-50,150-400 gpurun_out/r3/run17_build_stats.csv | head -6
./tools/ubench > gpurun_out/r3/run17_ubench.txt 2>&1; tail -24 gpurun_out/r3/run17_ubench.txt
kill $TICK
