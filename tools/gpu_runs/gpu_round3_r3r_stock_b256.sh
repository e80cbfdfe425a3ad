# Stock PyTorch-ROCm op path at the bench's default per-GPU batch, MIOpen's default (exhaustive) find mode:
# the equal-batch record of bench_baseline.json.  The search alone takes > 10 minutes; a heartbeat keeps the call alive.
mkdir -p gpurun_out/r3r
(while true; do sleep 60; date >> gpurun_out/r3r/heartbeat.log; done) &
HB=$!
timeout -k 10 1150 python bench.py --impl stock --steps 4 --warmup 2 --batch 256 > gpurun_out/r3r/stock_b256.log 2>&1
RC=$?
kill $HB
tail -2 gpurun_out/r3r/stock_b256.log | cut -c1-300
exit $RC
