mkdir -p gpurun_out/r3o && cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && export DPAMD_CONV_TUNE_FILE=$GRAFT_REPO_ROOT/gpurun_out/r3o/tune.json && \
timeout -k 10 200 python bench.py --steps 6 --warmup 4 > gpurun_out/r3o/t_b256.log 2>&1 && \
timeout -k 10 200 python bench.py --steps 6 --warmup 4 --batch 128 > gpurun_out/r3o/t_b128.log 2>&1 && \
timeout -k 10 200 python bench.py --model deeplabv3plus_xception --output-stride 8 --batch 64 --steps 6 --warmup 4 > gpurun_out/r3o/t_x65_b64.log 2>&1 && \
timeout -k 10 200 python bench.py --model deeplabv3plus_xception --output-stride 8 --batch 128 --steps 6 --warmup 4 > gpurun_out/r3o/t_x65_b128.log 2>&1 && \
timeout -k 10 200 python bench.py --model danet_resnet101 --task instance --crop 512 --batch 32 --steps 6 --warmup 4 > gpurun_out/r3o/t_danet_b32.log 2>&1 && \
DPAMD_DDP_FORCE=1 timeout -k 10 200 python bench.py --steps 6 --warmup 4 --sync-bn > gpurun_out/r3o/t_b256_syncbn_forced.log 2>&1 && \
timeout -k 10 200 python bench.py --steps 20 --warmup 8 > gpurun_out/r3o/b256.log 2>&1 && \
timeout -k 10 200 python bench.py --model danet_resnet101 --task instance --crop 512 --batch 32 --steps 20 --warmup 6 > gpurun_out/r3o/danet_b32.log 2>&1 && \
timeout -k 10 200 python bench.py --model deeplabv3plus_xception --output-stride 8 --batch 64 --steps 20 --warmup 6 > gpurun_out/r3o/x65_b64.log 2>&1 && \
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r3o/prof_b256 -o p -- python3 bench.py --steps 4 --warmup 3 > gpurun_out/r3o/prof_b256.log 2>&1 && \
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r3o/prof_danet -o p -- python3 bench.py --model danet_resnet101 --task instance --crop 512 --batch 32 --steps 4 --warmup 3 > gpurun_out/r3o/prof_danet.log 2>&1 && \
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r3o/prof_x65 -o p -- python3 bench.py --model deeplabv3plus_xception --output-stride 8 --batch 64 --steps 4 --warmup 3 > gpurun_out/r3o/prof_x65.log 2>&1 ; rm -f gpurun_out/r3o/*/*kernel_trace.csv; grep -o "\"value\": [0-9.]*" gpurun_out/r3o/*.log
