# edge: fewer validation samples than ranks on the device validation path (4 ranks sharing the GPU), both tasks
mkdir -p gpurun_out/r5s
run() { name=$1; shift; DPAMD_DIST_BACKEND=gloo timeout -k 10 250 "$@" > gpurun_out/r5s/$name.log 2>&1 || { echo "FAILED $name"; tail -25 gpurun_out/r5s/$name.log | cut -c1-200; exit 1; }; echo "ok $name: $(grep 'val:' gpurun_out/r5s/$name.log | tail -1 | cut -c1-170)"; }
run semseg_4ranks_tiny_val python -u train_pascal.py --synthetic --model deeplabv3plus_resnet18 --crop 129 --trainBatch 2 --testBatch 2 --synthetic-len 12 --nEpochs 1 --workers 0 --log-every 0 --hang-timeout 100 --gpus 4 --save-dir /tmp/s1
run instance_4ranks_tiny_val python -u train_pascal.py --synthetic --task instance --model danet_resnet50 --crop 128 --trainBatch 2 --testBatch 2 --synthetic-len 12 --nEpochs 1 --workers 0 --log-every 0 --hang-timeout 100 --gpus 4 --save-dir /tmp/s2
