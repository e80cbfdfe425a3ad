#!/bin/bash
# round-2 first GPU pass: tests, headline bench, stock baselines at equal batch, entrypoint throughput
set -o pipefail
mkdir -p gpurun_out/a
export TMPDIR=/tmp
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread 2>&1 | tee gpurun_out/a/pytest.log | grep -v "^$" | tail -8
timeout -k 10 300 python bench.py --steps 20 --warmup 5 --phases > gpurun_out/a/bench_b128.log 2>&1 || exit 1
tail -2 gpurun_out/a/bench_b128.log
timeout -k 10 300 python bench.py --impl stock --batch 64 --steps 10 --warmup 3 --record-baseline --source-log profiles/r2_stock_b64_bench.log > gpurun_out/a/stock_b64.log 2>&1 || exit 1
tail -1 gpurun_out/a/stock_b64.log
timeout -k 10 400 python bench.py --impl stock --batch 128 --steps 10 --warmup 3 --record-baseline --source-log profiles/r2_stock_b128_bench.log > gpurun_out/a/stock_b128.log 2>&1 || exit 1
tail -1 gpurun_out/a/stock_b128.log
cp bench_baseline.json gpurun_out/a/
timeout -k 10 400 python train_pascal.py --synthetic --model deeplabv3plus_resnet101 --crop 513 --trainBatch 128 --synthetic-len 1024 --nEpochs 3 --useTest 0 --workers 12 --save-dir /tmp/tp --log-every 4 > gpurun_out/a/train_pascal_b128.log 2>&1
echo "train_pascal rc=$?"
tail -6 gpurun_out/a/train_pascal_b128.log
