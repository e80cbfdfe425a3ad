# soak of the other two families through the entrypoint (normal-priority weight-gradient stream): Xception-65 OS8 B128 and
# DANet-R101 B64, ~500 steps each with device validation every 10 epochs, hang watchdog armed
mkdir -p gpurun_out/r7t
timeout -k 10 500 python -u train_pascal.py --synthetic --model deeplabv3plus_xception65 --output-stride 8 --crop 513 --trainBatch 128 --testBatch 64 --synthetic-len 2048 --nEpochs 30 --useTest 1 --nTestInterval 10 --workers 12 --save-dir /tmp/soakx2 --log-every 0 --hang-timeout 120 > gpurun_out/r7t/soak_x65.log 2>&1 || { tail -20 gpurun_out/r7t/soak_x65.log; exit 1; }
timeout -k 10 500 python -u train_pascal.py --synthetic --task instance --model danet_resnet101 --trainBatch 64 --testBatch 32 --synthetic-len 2048 --nEpochs 16 --useTest 1 --nTestInterval 8 --workers 12 --save-dir /tmp/soakd2 --log-every 0 --hang-timeout 120 > gpurun_out/r7t/soak_danet.log 2>&1 || { tail -20 gpurun_out/r7t/soak_danet.log; exit 1; }
grep "epoch:" gpurun_out/r7t/soak_x65.log | awk 'NR%5==1' | cut -c1-150; grep "val:" gpurun_out/r7t/soak_x65.log | cut -c1-180
grep "epoch:" gpurun_out/r7t/soak_danet.log | awk 'NR%3==1' | cut -c1-150; grep "val:" gpurun_out/r7t/soak_danet.log | cut -c1-200
