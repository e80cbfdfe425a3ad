#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/a
export TMPDIR=/tmp DPAMD_HEARTBEAT_S=45
timeout -k 10 500 python bench.py --impl stock --batch 64 --steps 10 --warmup 3 --record-baseline --source-log profiles/r2_stock_b64_bench.log 2>&1 | tee gpurun_out/a/stock_b64.log | tail -3 || exit 1
timeout -k 10 500 python bench.py --impl stock --batch 128 --steps 10 --warmup 3 --record-baseline --source-log profiles/r2_stock_b128_bench.log 2>&1 | tee gpurun_out/a/stock_b128.log | tail -3 || exit 1
cp bench_baseline.json gpurun_out/a/
timeout -k 10 400 python -u train_pascal.py --synthetic --model deeplabv3plus_resnet101 --crop 513 --trainBatch 128 --synthetic-len 1024 --nEpochs 3 --useTest 0 --workers 12 --save-dir /tmp/tp --log-every 4 2>&1 | tee gpurun_out/a/train_pascal_b128.log | tail -12
