# same-box A/B: cap of the statistics replicas the conv epilogues add into (4 / 8 = default / 16 / 32)
mkdir -p gpurun_out/r5k
for rep in 1 2; do for m in reps8 reps16 reps32 reps4; do
  timeout -k 10 300 python tools/ab_loader_sync.py $m --steps 16 --warmup 6 > gpurun_out/r5k/r101_${m}_${rep}.log 2>&1 || { tail -5 gpurun_out/r5k/r101_${m}_${rep}.log; exit 1; }
done; done
for m in reps8 reps16 reps32; do
  timeout -k 10 300 python tools/ab_loader_sync.py $m --model deeplabv3plus_xception65 --output-stride 8 --steps 10 --warmup 4 > gpurun_out/r5k/x65_${m}.log 2>&1 || exit 1
done
grep -o '"value": [0-9.]*' gpurun_out/r5k/*.log
