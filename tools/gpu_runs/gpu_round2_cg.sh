#!/bin/bash
# stock PyTorch-ROCm op path at per-GPU batch 256 (equal-batch record for bench.py's default); MIOpen's search may outlast the call
set -o pipefail
mkdir -p gpurun_out/cg
export TMPDIR=/tmp DPAMD_HEARTBEAT_S=45
timeout -k 10 1130 python bench.py --impl stock --batch 256 --steps 4 --warmup 1 --record-baseline --source-log profiles/r2_stock_b256_bench.log > gpurun_out/cg/stock_b256.log 2>&1
echo "rc=$?"; tail -2 gpurun_out/cg/stock_b256.log | cut -c1-400; cp bench_baseline.json gpurun_out/cg/bench_baseline.json
