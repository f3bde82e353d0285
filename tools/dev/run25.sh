set -x
mkdir -p gpurun_out/r3
( while sleep 45; do echo "tick $(date +%T)"; done ) &
TICK=$!
export TMPDIR=/tmp
for t in 0 4 8; do
  MIEKKI_TUNE_BUILD=$t timeout -k 10 200 rocprofv3 --kernel-trace --stats -d gpurun_out/r3/b25 -o d -- python3 tools/build_rate.py 6400 20 > gpurun_out/r3/run25_rate_t$t.txt 2>&1
  python tools/rocpd_stats.py gpurun_out/r3/b25/d_results.db > gpurun_out/r3/run25_stats_t$t.csv 2>&1; rm -rf gpurun_out/r3/b25
  tail -1 gpurun_out/r3/run25_rate_t$t.txt; grep scatter gpurun_out/r3/run25_stats_t$t.csv | cut -c1-40,150-300
done
kill $TICK
