# graph-step across the model families and through the entrypoint (poly schedule, validation between replays, snapshots)
mkdir -p gpurun_out/r6c
run() { name=$1; shift; timeout -k 10 300 "$@" > gpurun_out/r6c/$name.log 2>&1 || { echo "FAILED $name"; grep -v Warning gpurun_out/r6c/$name.log | tail -12 | cut -c1-300; exit 1; }; echo "ok $name $(grep -o '"value": [0-9.]*\|"final_loss": [0-9.e+]*\|"graphs_captured": [0-9]*' gpurun_out/r6c/$name.log | tr '\n' ' ')"; }
run x65_b8_eager python bench.py --model deeplabv3plus_xception65 --output-stride 8 --batch 8 --steps 20 --warmup 8
run x65_b8_graph python bench.py --model deeplabv3plus_xception65 --output-stride 8 --batch 8 --steps 20 --warmup 8 --graph-step
run danet_b8_eager python bench.py --model danet_resnet101 --task instance --crop 512 --batch 8 --steps 20 --warmup 8
run danet_b8_graph python bench.py --model danet_resnet101 --task instance --crop 512 --batch 8 --steps 20 --warmup 8 --graph-step
run fcn18_b4_graph python bench.py --model fcn_resnet18 --batch 4 --crop 321 --steps 20 --warmup 8 --graph-step
C="--synthetic --model deeplabv3plus_resnet101 --crop 513 --trainBatch 16 --testBatch 16 --synthetic-len 512 --nEpochs 6 --useTest 1 --nTestInterval 2 --snapshot 3 --workers 8 --log-every 0 --hang-timeout 120"
timeout -k 10 300 python -u train_pascal.py $C --save-dir /tmp/ge > gpurun_out/r6c/entry_eager.log 2>&1 || { tail -15 gpurun_out/r6c/entry_eager.log; exit 1; }
timeout -k 10 300 python -u train_pascal.py $C --graph-step --save-dir /tmp/gg > gpurun_out/r6c/entry_graph.log 2>&1 || { tail -15 gpurun_out/r6c/entry_graph.log; exit 1; }
echo "-- entrypoint eager"; grep "epoch:\|val:" gpurun_out/r6c/entry_eager.log | sed 's/numImages=[0-9]* //; s/host_rss.*//' | cut -c1-170
echo "-- entrypoint graph-step"; grep "epoch:\|val:" gpurun_out/r6c/entry_graph.log | sed 's/numImages=[0-9]* //; s/host_rss.*//' | cut -c1-170
