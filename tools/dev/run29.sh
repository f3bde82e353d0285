set -x
mkdir -p gpurun_out/r3
( while sleep 45; do echo "tick $(date +%T)"; done ) &
TICK=$!
export TMPDIR=/tmp
timeout -k 10 400 python -m pytest tests/test_gpu_packed.py tests/test_gpu_edges.py -x -q -m gpu > gpurun_out/r3/run29_pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r3/run29_pytest.log
tail -3 gpurun_out/r3/run29_pytest.log
for rep in 1 2; do timeout -k 10 100 python tools/build_rate.py 6400 20 > gpurun_out/r3/run29_rate_un2_$rep.txt 2>&1; tail -1 gpurun_out/r3/run29_rate_un2_$rep.txt; done
timeout -k 10 200 rocprofv3 --kernel-trace --stats -d gpurun_out/r3/b29 -o d -- python3 tools/build_rate.py 6400 20 > gpurun_out/r3/b29.log 2>&1
python tools/rocpd_stats.py gpurun_out/r3/b29/d_results.db > gpurun_out/r3/run29_build_stats.csv 2>&1
python tools/rocpd_timeline.py gpurun_out/r3/b29/d_results.db 900 60 > gpurun_out/r3/run29_timeline.txt 2>&1
rm -rf gpurun_out/r3/b29
cut -c1-50,150-400 gpurun_out/r3/run29_build_stats.csv | head -12
cp miekki_amd/libmiekki_hip.so /tmp/lib_un2.so
cp miekki_amd/variants/libmiekki_hip_un3.so miekki_amd/libmiekki_hip.so
for rep in 1 2; do timeout -k 10 100 python tools/build_rate.py 6400 20 > gpurun_out/r3/run29_rate_un3_$rep.txt 2>&1; tail -1 gpurun_out/r3/run29_rate_un3_$rep.txt; done
cp /tmp/lib_un2.so miekki_amd/libmiekki_hip.so
kill $TICK
