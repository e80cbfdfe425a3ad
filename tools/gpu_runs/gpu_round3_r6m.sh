# instance-task validation at testBatch = 1 / 4: eager forward against hipGraph replay; Jaccard must not move
mkdir -p gpurun_out/r6m
C="--synthetic --task instance --model danet_resnet101 --trainBatch 8 --synthetic-len 256 --nEpochs 1 --useTest 1 --workers 8 --log-every 0 --hang-timeout 120 --max-iters 4 --seed 1"
run() { name=$1; shift; DPAMD_DETERMINISTIC=1 timeout -k 10 300 python -u train_pascal.py $C "$@" > gpurun_out/r6m/$name.log 2>&1 || { echo "FAILED $name"; grep -v Warning gpurun_out/r6m/$name.log | tail -10 | cut -c1-250; exit 1; }; echo "$name: $(grep 'val:' gpurun_out/r6m/$name.log | tail -1 | cut -c1-200)"; }
run b1_eager --testBatch 1 --graph-eval off --save-dir /tmp/m1
run b1_graph --testBatch 1 --graph-eval on --save-dir /tmp/m2
run b4_eager --testBatch 4 --graph-eval off --save-dir /tmp/m3
run b4_graph --testBatch 4 --graph-eval on --save-dir /tmp/m4
timeout -k 10 300 python -u -m pytest tests/test_gpu_runtime.py -x -q --timeout 300 --timeout-method thread -k "entrypoint" > gpurun_out/r6m/pytest.log 2>&1; echo "pytest rc=$? $(tail -1 gpurun_out/r6m/pytest.log)"
