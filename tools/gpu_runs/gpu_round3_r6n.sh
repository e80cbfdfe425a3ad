# instance-task device validation at testBatch = 1, second epoch (stores built, graphs captured): eager vs replay
mkdir -p gpurun_out/r6n
C="--synthetic --task instance --model danet_resnet101 --trainBatch 8 --testBatch 1 --synthetic-len 512 --nEpochs 2 --useTest 1 --workers 8 --log-every 0 --hang-timeout 120 --max-iters 40"
for m in off on; do timeout -k 10 300 python -u train_pascal.py $C --graph-eval $m --save-dir /tmp/n_$m > gpurun_out/r6n/graph_$m.log 2>&1 || { tail -8 gpurun_out/r6n/graph_$m.log; exit 1; }; echo "graph-eval $m: $(grep 'val:' gpurun_out/r6n/graph_$m.log | tail -1 | sed 's/JACCARD=\[[^]]*\] //' | cut -c1-160)"; done
