# memory over many epochs (host RSS, device allocated / reserved in the epoch log): R101 B128, 40 epochs of 8 steps with
# device validation and snapshots every 10, stream-mode loader in a second run
mkdir -p gpurun_out/r5w
C="--synthetic --model deeplabv3plus_resnet101 --crop 513 --trainBatch 128 --testBatch 64 --synthetic-len 1024 --useTest 1 --nTestInterval 5 --snapshot 10 --workers 8 --log-every 0 --hang-timeout 120"
timeout -k 10 500 python -u train_pascal.py $C --nEpochs 40 --save-dir /tmp/mem1 > gpurun_out/r5w/cache_mode.log 2>&1 || { tail -20 gpurun_out/r5w/cache_mode.log; exit 1; }
timeout -k 10 500 python -u train_pascal.py $C --nEpochs 12 --gpu-cache-gb 0.001 --save-dir /tmp/mem2 > gpurun_out/r5w/stream_mode.log 2>&1 || { tail -20 gpurun_out/r5w/stream_mode.log; exit 1; }
grep "epoch:" gpurun_out/r5w/cache_mode.log | awk 'NR%5==1' | cut -c1-230; echo "-- stream mode"; grep "epoch:" gpurun_out/r5w/stream_mode.log | awk 'NR%2==1' | cut -c1-230
