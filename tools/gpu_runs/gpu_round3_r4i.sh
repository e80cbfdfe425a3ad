#!/bin/bash
# the product entrypoint on the device data path, final tree: semseg R101 B256 (3 epochs of 2048 synthetic images, device
# validation each epoch), instance task DANet-R101 B64 (2 epochs of 1024, device validation), Xception-65 OS8 B128
mkdir -p gpurun_out/r4i
timeout -k 10 400 python -u train_pascal.py --synthetic --model deeplabv3plus_resnet101 --crop 513 --trainBatch 256 --testBatch 64 --synthetic-len 2048 --nEpochs 3 --useTest 1 --workers 12 --save-dir /tmp/tp256 --log-every 4 > gpurun_out/r4i/train_pascal_r101_b256.log 2>&1 || { tail -20 gpurun_out/r4i/train_pascal_r101_b256.log; exit 1; }
timeout -k 10 400 python -u train_pascal.py --synthetic --task instance --model danet_resnet101 --trainBatch 64 --testBatch 32 --synthetic-len 1024 --nEpochs 2 --useTest 1 --workers 12 --save-dir /tmp/tpdanet --log-every 4 > gpurun_out/r4i/train_pascal_danet_b64.log 2>&1 || { tail -20 gpurun_out/r4i/train_pascal_danet_b64.log; exit 1; }
timeout -k 10 400 python -u train_pascal.py --synthetic --model deeplabv3plus_xception --output-stride 8 --crop 513 --trainBatch 128 --testBatch 64 --synthetic-len 1024 --nEpochs 2 --useTest 1 --workers 12 --save-dir /tmp/tpx65 --log-every 2 > gpurun_out/r4i/train_pascal_x65_b128.log 2>&1 || { tail -20 gpurun_out/r4i/train_pascal_x65_b128.log; exit 1; }
grep -h "epoch:\|val" gpurun_out/r4i/*.log | cut -c1-260
