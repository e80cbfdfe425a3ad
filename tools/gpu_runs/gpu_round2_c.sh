#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/c
export TMPDIR=/tmp DPAMD_HEARTBEAT_S=45
timeout -k 10 300 python -u -m pytest tests/test_gpu_attention.py -m gpu -x -q --timeout 200 --timeout-method thread 2>&1 | tee gpurun_out/c/pytest_att.log | tail -3 || exit 1
timeout -k 10 400 python bench.py --task instance --model danet_resnet101 --crop 512 --batch 16 --steps 10 --warmup 4 --phases 2>&1 | tee gpurun_out/c/danet_b16.log | tail -3 || exit 1
timeout -k 10 400 python bench.py --task instance --model danet_resnet101 --crop 512 --batch 32 --steps 10 --warmup 3 --phases 2>&1 | tee gpurun_out/c/danet_b32.log | tail -3 || exit 1
cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/c/prof -- python $GRAFT_REPO_ROOT/bench.py --task instance --model danet_resnet101 --crop 512 --batch 16 --steps 4 --warmup 2 > $GRAFT_REPO_ROOT/gpurun_out/c/prof.log 2>&1
cd $GRAFT_REPO_ROOT && python tools/summarize_prof.py $(find gpurun_out/c/prof -name "*kernel_stats.csv" | head -1) 6 "DANet-R101 512 B16 instance task (2 warmup + 4 steps)" > gpurun_out/c/danet_b16_profile.md 2>&1; head -45 gpurun_out/c/danet_b16_profile.md
timeout -k 10 300 python -u train_pascal.py --task instance --synthetic --model danet_resnet101 --trainBatch 16 --synthetic-len 64 --nEpochs 2 --workers 8 --save-dir /tmp/tpi --log-every 2 --testBatch 1 --max-iters 4 2>&1 | tee gpurun_out/c/train_pascal_instance.log | tail -8
