# the small-batch regime (the reference's own default is trainBatch 16): R101 513^2 at B8 / B16 / B32 / B64, with the
# phase timers (host-side launch time against device time)
mkdir -p gpurun_out/r5z
for b in 8 16 32 64; do
  timeout -k 10 300 python bench.py --batch $b --steps 30 --warmup 10 --phases > gpurun_out/r5z/r101_b$b.log 2>&1 || { tail -5 gpurun_out/r5z/r101_b$b.log; exit 1; }
done
grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*\|"tiles_autotuned_in_run": [0-9]*' gpurun_out/r5z/*.log; grep -h "phase\|fwd\|bwd" gpurun_out/r5z/r101_b16.log | grep -v "^{" | head -8 | cut -c1-200
