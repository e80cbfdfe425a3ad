# same-box A/B: DANet B64 / B32 with and without a per-batch device synchronisation in the loader (interleaved, twice)
mkdir -p gpurun_out/r4l
A="--model danet_resnet101 --task instance --crop 512 --steps 20 --warmup 6"
for rep in 1 2; do for m in nosync sync; do for b in 64 32; do
  timeout -k 10 300 python tools/ab_loader_sync.py $m $A --batch $b > gpurun_out/r4l/danet_b${b}_${m}_${rep}.log 2>&1 || { tail -5 gpurun_out/r4l/danet_b${b}_${m}_${rep}.log; exit 1; }
done; done; done
grep -o '"value": [0-9.]*\|"hbm_reserved_peak_gib": [0-9.]*' gpurun_out/r4l/*.log
