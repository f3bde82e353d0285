set -x
mkdir -p gpurun_out/r3
( while sleep 45; do echo "tick $(date +%T)"; done ) &
TICK=$!
export TMPDIR=/tmp
timeout -k 10 400 python -m pytest tests/test_gpu_packed.py tests/test_gpu_edges.py -x -q -m gpu > gpurun_out/r3/run26_pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r3/run26_pytest.log
tail -12 gpurun_out/r3/run26_pytest.log
for t in 0 1 2; do
  MIEKKI_TUNE_BUILD=$t timeout -k 10 100 python tools/build_rate.py 6400 20 > gpurun_out/r3/run26_rate_t$t.txt 2>&1; tail -1 gpurun_out/r3/run26_rate_t$t.txt
done
timeout -k 10 200 rocprofv3 --kernel-trace --stats -d gpurun_out/r3/b26 -o d -- python3 tools/build_rate.py 6400 20 > gpurun_out/r3/b26.log 2>&1
python tools/rocpd_stats.py gpurun_out/r3/b26/d_results.db > gpurun_out/r3/run26_build_stats.csv 2>&1
python - <<'PY' > gpurun_out/r3/run26_reduce_by_batch.txt 2>&1
import sqlite3
db=sqlite3.connect('gpurun_out/r3/b26/d_results.db'); cur=db.cursor()
sym=[r[1] for r in cur.execute("pragma table_info(rocpd_info_kernel_symbol)")]
nc="kernel_name" if "kernel_name" in sym else "display_name"
for k in ("build_reduce","build_scatter"):
    rows=list(cur.execute(f"select d.end-d.start from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s on d.kernel_id=s.id where s.{nc} like '%{k}%' order by d.start"))
    print(k, "duration (us) by batch:", " ".join(f"{r[0]/1e3:.0f}" for r in rows))
PY
rm -rf gpurun_out/r3/b26
cat gpurun_out/r3/run26_reduce_by_batch.txt
cut -c1-50,150-400 gpurun_out/r3/run26_build_stats.csv | head -4
kill $TICK
