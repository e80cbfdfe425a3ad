# soak: the entrypoint for ~1,000 steps (R101 B256, 60 epochs of 4,096 synthetic images, device validation every 10 epochs,
# snapshots every 20, hang watchdog armed), then a resume for 4 more epochs; throughput and memory per epoch
mkdir -p gpurun_out/r6z
C="--synthetic --model deeplabv3plus_resnet101 --crop 513 --trainBatch 256 --testBatch 64 --synthetic-len 4096 --useTest 1 --nTestInterval 10 --snapshot 20 --workers 12 --save-dir /tmp/soak2 --log-every 0 --hang-timeout 120"
timeout -k 10 700 python -u train_pascal.py $C --nEpochs 60 > gpurun_out/r6z/soak.log 2>&1 || { tail -20 gpurun_out/r6z/soak.log; exit 1; }
timeout -k 10 300 python -u train_pascal.py $C --nEpochs 64 --resume_epoch 60 > gpurun_out/r6z/resume.log 2>&1 || { tail -20 gpurun_out/r6z/resume.log; exit 1; }
grep "epoch:" gpurun_out/r6z/soak.log | awk 'NR%6==1' | cut -c1-160; grep "val:" gpurun_out/r6z/soak.log | cut -c1-200; grep "Resumed\|epoch:\|val:" gpurun_out/r6z/resume.log | cut -c1-200
