# host-device synchronisations in steady-state iterations of the three model families (tools/find_syncs.py)
mkdir -p gpurun_out/r4m
C="--synthetic --nEpochs 1 --useTest 0 --log-every 0 --max-iters 10 --workers 8 --save-dir /tmp/fs"
timeout -k 10 300 python -u tools/find_syncs.py -- $C --model deeplabv3plus_resnet101 --trainBatch 64 --synthetic-len 1024 > gpurun_out/r4m/r101.log 2>&1 && \
timeout -k 10 300 python -u tools/find_syncs.py -- $C --task instance --model danet_resnet101 --trainBatch 32 --synthetic-len 512 > gpurun_out/r4m/danet.log 2>&1 && \
timeout -k 10 300 python -u tools/find_syncs.py -- $C --model deeplabv3plus_xception --output-stride 8 --trainBatch 32 --synthetic-len 512 > gpurun_out/r4m/x65.log 2>&1; rc=$?
for f in r101 danet x65; do echo "== $f"; sed -n '/synchronising calls/,$p' gpurun_out/r4m/$f.log | cut -c1-400; done; exit $rc
