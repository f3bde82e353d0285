set -x
mkdir -p gpurun_out/r3
( while sleep 45; do echo "tick $(date +%T)"; done ) &
TICK=$!
export TMPDIR=/tmp
for n in 1 2 3 4; do
  MIEKKI_COPY_STREAMS=$n timeout -k 10 120 python tools/host_fed_rate.py 24 > gpurun_out/r3/run22_hostfed_$n.txt 2>&1; tail -2 gpurun_out/r3/run22_hostfed_$n.txt
done
kill $TICK
