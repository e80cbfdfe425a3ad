# stream mode after the copy stream became per-loader: reserved memory over 16 epochs
mkdir -p gpurun_out/r5x
C="--synthetic --model deeplabv3plus_resnet101 --crop 513 --trainBatch 128 --testBatch 64 --synthetic-len 1024 --useTest 1 --nTestInterval 5 --snapshot 10 --workers 8 --log-every 0 --hang-timeout 120"
timeout -k 10 500 python -u train_pascal.py $C --nEpochs 16 --gpu-cache-gb 0.001 --save-dir /tmp/mem3 > gpurun_out/r5x/stream_mode.log 2>&1 || { tail -20 gpurun_out/r5x/stream_mode.log; exit 1; }
grep "epoch:" gpurun_out/r5x/stream_mode.log | awk 'NR%2==1' | cut -c1-230
