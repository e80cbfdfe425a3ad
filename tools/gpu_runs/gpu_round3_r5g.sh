# host-device synchronisations through validation and in stream mode (tools/find_syncs.py)
mkdir -p gpurun_out/r5g
C="--synthetic --nEpochs 1 --useTest 1 --log-every 0 --max-iters 8 --workers 4 --snapshot 100"
timeout -k 10 300 python -u tools/find_syncs.py -- $C --model deeplabv3plus_resnet18 --crop 257 --trainBatch 16 --testBatch 16 --synthetic-len 256 --save-dir /tmp/g1 > gpurun_out/r5g/semseg_val.log 2>&1 && \
timeout -k 10 300 python -u tools/find_syncs.py -- $C --task instance --model danet_resnet50 --crop 256 --trainBatch 16 --testBatch 16 --synthetic-len 256 --save-dir /tmp/g2 > gpurun_out/r5g/instance_val.log 2>&1 && \
timeout -k 10 300 python -u tools/find_syncs.py -- $C --model deeplabv3plus_resnet18 --crop 257 --trainBatch 16 --testBatch 16 --synthetic-len 256 --gpu-cache-gb 0.0001 --save-dir /tmp/g3 > gpurun_out/r5g/semseg_stream.log 2>&1; rc=$?
for f in semseg_val instance_val semseg_stream; do echo "== $f"; sed -n '/synchronising calls/,$p' gpurun_out/r5g/$f.log | cut -c1-330; done; exit $rc
