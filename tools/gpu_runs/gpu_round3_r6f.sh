# graph-step next to loaders that keep background threads busy (pin-memory thread of the host data path, stream mode),
# with frozen BN, with SyncBN requested on one rank, resume into graph-step, and refused configurations
mkdir -p gpurun_out/r6f
B="--synthetic --model deeplabv3plus_resnet50 --crop 257 --trainBatch 8 --testBatch 8 --synthetic-len 128 --nEpochs 3 --workers 4 --log-every 0 --hang-timeout 100 --graph-step"
run() { name=$1; shift; timeout -k 10 250 "$@" > gpurun_out/r6f/$name.log 2>&1 || { echo "FAILED $name"; grep -v Warning gpurun_out/r6f/$name.log | tail -12 | cut -c1-260; exit 1; }; echo "ok $name: $(grep 'val:' gpurun_out/r6f/$name.log | tail -1 | cut -c1-130)"; }
run host_data python -u train_pascal.py $B --data-path cpu --save-dir /tmp/h1
run stream_mode python -u train_pascal.py $B --gpu-cache-gb 0.0001 --save-dir /tmp/h2
run freeze_bn python -u train_pascal.py $B --freeze-bn --lr 1e-6 --save-dir /tmp/h3
run syncbn_1rank python -u train_pascal.py $B --sync-bn --save-dir /tmp/h4
run snapshot python -u train_pascal.py $B --snapshot 2 --save-dir /tmp/h5
run resume python -u train_pascal.py $B --nEpochs 4 --resume_epoch 2 --save-dir /tmp/h5
run instance python -u train_pascal.py --synthetic --task instance --model danet_resnet50 --crop 128 --trainBatch 8 --testBatch 8 --synthetic-len 64 --nEpochs 3 --workers 4 --log-every 0 --graph-step --save-dir /tmp/h6
timeout -k 10 100 python -u train_pascal.py $B --nAveGrad 2 --save-dir /tmp/h7 > gpurun_out/r6f/refused_accum.log 2>&1; echo "accumulation with graph-step: rc=$? $(grep -o 'NotImplementedError.*' gpurun_out/r6f/refused_accum.log | head -1 | cut -c1-120)"
