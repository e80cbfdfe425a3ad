# more ranks and more flags on the shared GPU (gloo): 4 ranks, reduce-scatter/all-gather buckets, accumulation, stream
# mode, host data path, resume under 2 ranks, debug checks
mkdir -p gpurun_out/r5c
B="--synthetic --model deeplabv3plus_resnet18 --crop 129 --trainBatch 4 --testBatch 4 --synthetic-len 72 --workers 2 --log-every 0 --hang-timeout 100"
run() { name=$1; shift; DPAMD_DIST_BACKEND=gloo timeout -k 10 250 "$@" > gpurun_out/r5c/$name.log 2>&1 || { echo "FAILED $name"; tail -25 gpurun_out/r5c/$name.log | cut -c1-200; exit 1; }; echo "ok $name: $(grep 'val:' gpurun_out/r5c/$name.log | tail -1 | cut -c1-140)"; }
run four_ranks python -u train_pascal.py $B --nEpochs 2 --gpus 4 --save-dir /tmp/c1
run four_ranks_syncbn python -u train_pascal.py $B --nEpochs 2 --gpus 4 --sync-bn --save-dir /tmp/c2
run accum2_2ranks python -u train_pascal.py $B --nEpochs 2 --gpus 2 --nAveGrad 2 --save-dir /tmp/c3
run stream_2ranks python -u train_pascal.py $B --nEpochs 2 --gpus 2 --gpu-cache-gb 0.0001 --save-dir /tmp/c4
run hostdata_2ranks python -u train_pascal.py $B --nEpochs 2 --gpus 2 --data-path cpu --save-dir /tmp/c5
run snapshot_2ranks python -u train_pascal.py $B --nEpochs 2 --gpus 2 --snapshot 2 --save-dir /tmp/c6
run resume_2ranks python -u train_pascal.py $B --nEpochs 3 --gpus 2 --resume_epoch 2 --save-dir /tmp/c6
run debug_2ranks env DPAMD_DEBUG=1 python -u train_pascal.py $B --nEpochs 2 --gpus 2 --save-dir /tmp/c7
run x65_3ranks python -u train_pascal.py $B --model deeplabv3plus_xception65 --output-stride 16 --nEpochs 1 --gpus 3 --save-dir /tmp/c8
ls /tmp/c6/run_0/models | head
