# flag matrix of the entrypoint on the GPU path (small model and crop, a few iterations each); stops at the first failure
mkdir -p gpurun_out/r4y
B="--synthetic --model deeplabv3plus_resnet18 --crop 129 --trainBatch 8 --testBatch 8 --synthetic-len 64 --nEpochs 2 --workers 4 --log-every 0 --hang-timeout 100"
run() { name=$1; shift; timeout -k 10 200 "$@" > gpurun_out/r4y/$name.log 2>&1 || { echo "FAILED $name"; tail -15 gpurun_out/r4y/$name.log; exit 1; }; echo "ok $name: $(grep 'val:' gpurun_out/r4y/$name.log | tail -1 | cut -c1-150)"; }
run stream_mode python -u train_pascal.py $B --save-dir /tmp/m1 --gpu-cache-gb 0.0001
run accum2 python -u train_pascal.py $B --save-dir /tmp/m2 --nAveGrad 2
run freeze_bn python -u train_pascal.py $B --save-dir /tmp/m3 --freeze-bn --lr 1e-6   # (random init without normalisation diverges at 0.007 on the fp32 CPU path too: the guard stops it)
run syncbn_forced env DPAMD_DDP_FORCE=1 python -u train_pascal.py $B --save-dir /tmp/m4 --sync-bn
run head_lr_cos python -u train_pascal.py $B --save-dir /tmp/m5 --head-lr-mult 10 --lr-mode cos
run cpu_data_path python -u train_pascal.py $B --save-dir /tmp/m6 --data-path cpu
run val_cpu python -u train_pascal.py $B --save-dir /tmp/m7 --val-path cpu
run xception_os16 python -u train_pascal.py $B --save-dir /tmp/m8 --model deeplabv3plus_xception --output-stride 16
run instance_r50 python -u train_pascal.py --synthetic --task instance --model danet_resnet50 --crop 128 --trainBatch 8 --testBatch 8 --synthetic-len 64 --nEpochs 2 --workers 4 --log-every 0 --save-dir /tmp/m9
run instance_stream python -u train_pascal.py --synthetic --task instance --model danet_resnet50 --crop 128 --trainBatch 8 --testBatch 8 --synthetic-len 64 --nEpochs 2 --workers 4 --log-every 0 --save-dir /tmp/m10 --gpu-cache-gb 0.0001
run deterministic env DPAMD_DETERMINISTIC=1 python -u train_pascal.py $B --save-dir /tmp/m11 --snapshot 2
run resume python -u train_pascal.py $B --save-dir /tmp/m11 --nEpochs 3 --resume_epoch 2
run debug_asserts env DPAMD_DEBUG=1 python -u train_pascal.py $B --save-dir /tmp/m12
run two_ranks_one_gpu env DPAMD_DIST_BACKEND=gloo python -u train_pascal.py $B --save-dir /tmp/m13 --gpus 2
