# validation at the reference's testBatch = 1: eager forward against hipGraph replay, device and host validation paths
mkdir -p gpurun_out/r6k
C="--synthetic --model deeplabv3plus_resnet101 --crop 513 --trainBatch 16 --testBatch 1 --synthetic-len 640 --nEpochs 2 --useTest 1 --workers 8 --log-every 0 --hang-timeout 120 --max-iters 6"
run() { name=$1; shift; timeout -k 10 300 python -u train_pascal.py $C "$@" > gpurun_out/r6k/$name.log 2>&1 || { echo "FAILED $name"; grep -v Warning gpurun_out/r6k/$name.log | tail -10 | cut -c1-250; exit 1; }; echo "$name: $(grep 'val:' gpurun_out/r6k/$name.log | tail -1 | cut -c1-170)"; }
run device_eager --graph-eval off --save-dir /tmp/k1
run device_graph --graph-eval on --save-dir /tmp/k2
run host_eager --val-path cpu --graph-eval off --save-dir /tmp/k3
run host_graph --val-path cpu --graph-eval on --save-dir /tmp/k4
