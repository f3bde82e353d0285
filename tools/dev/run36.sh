mkdir -p gpurun_out/r3
for sp in 16 64 256; do
  for n in 1024 12800; do
    MIEKKI_PASSB_SPAN=$sp timeout -k 10 100 python tools/build_rate.py $n 20 > gpurun_out/r3/run36_$sp_$n.txt 2>&1; echo "span $sp n $n: $(tail -1 gpurun_out/r3/run36_$sp_$n.txt)"
  done
done
