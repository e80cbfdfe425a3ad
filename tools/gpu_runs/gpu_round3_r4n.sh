mkdir -p gpurun_out/r4n
timeout -k 10 300 python -u -m pytest tests/test_gpu_runtime.py -x -q -s --timeout 120 --timeout-method thread -k "checkpoint or entrypoint or loader" > gpurun_out/r4n/pytest.log 2>&1 && \
timeout -k 10 300 python -u tools/find_syncs.py -- --synthetic --nEpochs 1 --useTest 0 --log-every 0 --max-iters 10 --workers 8 --save-dir /tmp/fs --model deeplabv3plus_resnet101 --trainBatch 64 --synthetic-len 1024 > gpurun_out/r4n/r101.log 2>&1; rc=$?
grep "arena positions" gpurun_out/r4n/pytest.log; tail -3 gpurun_out/r4n/pytest.log; sed -n '/synchronising calls/,$p' gpurun_out/r4n/r101.log | cut -c1-300; exit $rc
