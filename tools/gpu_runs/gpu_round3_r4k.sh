# after the blocking identity-matrix copy left the instance loader: entrypoint log (data_wait_ms, gpu_busy_at_fetch) and the
# DANet bench at B32 / B64; GPU tests of the data path
mkdir -p gpurun_out/r4k && \
timeout -k 10 300 python -u -m pytest tests/test_gpu_runtime.py tests/test_gpu_attention.py -x -q --timeout 120 --timeout-method thread > gpurun_out/r4k/pytest.log 2>&1 && \
timeout -k 10 400 python -u train_pascal.py --synthetic --task instance --model danet_resnet101 --trainBatch 64 --testBatch 32 --synthetic-len 1024 --nEpochs 2 --useTest 1 --workers 12 --save-dir /tmp/tpdanet --log-every 4 > gpurun_out/r4k/train_pascal_danet_b64.log 2>&1 && \
timeout -k 10 300 python bench.py --model danet_resnet101 --task instance --crop 512 --batch 64 --steps 20 --warmup 6 > gpurun_out/r4k/danet_b64.log 2>&1 && \
timeout -k 10 300 python bench.py --model danet_resnet101 --task instance --crop 512 --batch 32 --steps 20 --warmup 6 > gpurun_out/r4k/danet_b32.log 2>&1; rc=$?
tail -2 gpurun_out/r4k/pytest.log; grep "train:\|epoch:\|val:" gpurun_out/r4k/train_pascal_danet_b64.log | cut -c1-300; grep -o '"value": [0-9.]*' gpurun_out/r4k/danet_b*.log; exit $rc
