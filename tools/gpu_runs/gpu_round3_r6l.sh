# after graph evaluation became automatic for testBatch <= 4: the entrypoint tests, the batch-1 validation numbers again,
# two ranks with batch-2 validation
mkdir -p gpurun_out/r6l
timeout -k 10 600 python -u -m pytest tests/test_gpu_runtime.py tests/test_gpu_p2p.py -x -q --timeout 300 --timeout-method thread > gpurun_out/r6l/pytest.log 2>&1 || { grep -n "^E  \|Error" gpurun_out/r6l/pytest.log | head -12 | cut -c1-300; tail -3 gpurun_out/r6l/pytest.log; exit 1; }
tail -1 gpurun_out/r6l/pytest.log
bash tools/gpu_runs/gpu_round3_r6k.sh || exit 1
DPAMD_DIST_BACKEND=gloo timeout -k 10 250 python -u train_pascal.py --synthetic --model deeplabv3plus_resnet18 --crop 129 --trainBatch 4 --testBatch 2 --synthetic-len 72 --nEpochs 2 --workers 2 --log-every 0 --hang-timeout 100 --gpus 2 --save-dir /tmp/l1 > gpurun_out/r6l/two_ranks.log 2>&1 || { tail -20 gpurun_out/r6l/two_ranks.log | cut -c1-200; exit 1; }
echo "two ranks: $(grep 'val:' gpurun_out/r6l/two_ranks.log | tail -1 | cut -c1-160)"
