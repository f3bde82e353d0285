set -x
mkdir -p gpurun_out/r3
( while sleep 45; do echo "tick $(date +%T)"; done ) &
TICK=$!
export TMPDIR=/tmp
MIEKKI_COPY_STREAMS=1 timeout -k 10 200 rocprofv3 --kernel-trace --memory-copy-trace -d gpurun_out/r3/hf -o d -- python3 tools/host_fed_rate.py 8 > gpurun_out/r3/hf.log 2>&1
ls gpurun_out/r3/hf
python - <<'PY' > gpurun_out/r3/run21_tables.txt 2>&1
import sqlite3,glob
db=sqlite3.connect(glob.glob('gpurun_out/r3/hf/*.db')[0]); cur=db.cursor()
for (n,) in list(cur.execute("select name from sqlite_master where name like '%memory%'")):
    print(n, [r[1] for r in cur.execute(f"pragma table_info({n})")])
PY
cat gpurun_out/r3/run21_tables.txt
python tools/rocpd_timeline.py gpurun_out/r3/hf/d_results.db 60 40 > gpurun_out/r3/run21_timeline.txt 2>&1
rm -rf gpurun_out/r3/hf
head -150 gpurun_out/r3/run21_timeline.txt
kill $TICK
