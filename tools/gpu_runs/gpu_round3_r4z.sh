# two ranks sharing the GPU (gloo) through the entrypoint, after tile agreement moved into collective sections:
# the master-only validation figure, per-rank validation tails, SyncBN, Xception-65, the instance task
mkdir -p gpurun_out/r4z
B="--synthetic --crop 129 --trainBatch 8 --testBatch 8 --synthetic-len 72 --nEpochs 2 --workers 2 --log-every 0 --hang-timeout 100 --gpus 2"
run() { name=$1; shift; DPAMD_DIST_BACKEND=gloo timeout -k 10 250 "$@" > gpurun_out/r4z/$name.log 2>&1 || { echo "FAILED $name"; tail -25 gpurun_out/r4z/$name.log | cut -c1-200; exit 1; }; echo "ok $name: $(grep 'val:' gpurun_out/r4z/$name.log | tail -1 | cut -c1-150)"; }
run r18 python -u train_pascal.py $B --model deeplabv3plus_resnet18 --save-dir /tmp/z1
run r18_syncbn python -u train_pascal.py $B --model deeplabv3plus_resnet18 --sync-bn --save-dir /tmp/z2
run x65_syncbn python -u train_pascal.py $B --model deeplabv3plus_xception --output-stride 16 --sync-bn --save-dir /tmp/z3
run instance_r50 python -u train_pascal.py --synthetic --task instance --model danet_resnet50 --crop 128 --trainBatch 8 --testBatch 8 --synthetic-len 72 --nEpochs 2 --workers 2 --log-every 0 --hang-timeout 100 --gpus 2 --save-dir /tmp/z4
run bench_2ranks python bench.py --gpus 2 --model deeplabv3plus_resnet50 --crop 257 --batch 8 --steps 4 --warmup 2
grep -o '"tiles_autotuned_in_run": [0-9]*\|"value": [0-9.]*' gpurun_out/r4z/bench_2ranks.log
