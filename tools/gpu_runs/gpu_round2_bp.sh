#!/bin/bash
# per-GPU batch beyond 128 on the batch-chunked second-generation kernels: B256 (x2, tile choices kept), B384, B512
set -o pipefail
mkdir -p gpurun_out/bp
export TMPDIR=/tmp DPAMD_HEARTBEAT_S=45
for i in 1 2 3; do
DPAMD_CONV_TUNE_FILE=$GRAFT_REPO_ROOT/gpurun_out/bp/r101_b256_tune_$i.json timeout -k 10 500 python bench.py --batch 256 --steps 10 --warmup 3 > gpurun_out/bp/r101_b256_$i.log 2>&1 || { tail -5 gpurun_out/bp/r101_b256_$i.log; exit 1; }
tail -1 gpurun_out/bp/r101_b256_$i.log | cut -c1-200
done
timeout -k 10 500 python bench.py --batch 384 --steps 6 --warmup 3 > gpurun_out/bp/r101_b384.log 2>&1; tail -1 gpurun_out/bp/r101_b384.log | cut -c1-200
rocm-smi --showmeminfo vram 2>/dev/null | grep -i "used" | head -2
