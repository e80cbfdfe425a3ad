#!/bin/bash
# the product entrypoint on the device data path, v38: train_pascal.py --synthetic at per-GPU batch 256 and 128
set -o pipefail
mkdir -p gpurun_out/ce
export TMPDIR=/tmp DPAMD_HEARTBEAT_S=45
timeout -k 10 500 python -u train_pascal.py --synthetic --model deeplabv3plus_resnet101 --crop 513 --trainBatch 256 --synthetic-len 2048 --nEpochs 3 --useTest 0 --workers 12 --save-dir /tmp/tp256 --log-every 4 > gpurun_out/ce/train_pascal_b256.log 2>&1 || { tail -20 gpurun_out/ce/train_pascal_b256.log; exit 1; }
grep "epoch:" gpurun_out/ce/train_pascal_b256.log | cut -c1-200
timeout -k 10 500 python -u train_pascal.py --synthetic --model deeplabv3plus_resnet101 --crop 513 --trainBatch 128 --synthetic-len 2048 --nEpochs 3 --useTest 0 --workers 12 --save-dir /tmp/tp128 --log-every 4 > gpurun_out/ce/train_pascal_b128.log 2>&1 || { tail -20 gpurun_out/ce/train_pascal_b128.log; exit 1; }
grep "epoch:" gpurun_out/ce/train_pascal_b128.log | cut -c1-200
