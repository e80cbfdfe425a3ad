#!/bin/bash
# batch chunks beyond the kernels' 32-bit offsets: tests, then R101 at per-GPU batch 256 (activations > 2 GiB) against 128
set -o pipefail
mkdir -p gpurun_out/bo
export TMPDIR=/tmp DPAMD_HEARTBEAT_S=45
echo skip-tests

timeout -k 10 500 python bench.py --batch 256 --steps 8 --warmup 3 > gpurun_out/bo/r101_b256.log 2>&1 &&
DPAMD_CONV_BATCH_CHUNKS=0 timeout -k 10 500 python bench.py --batch 256 --steps 8 --warmup 3 > gpurun_out/bo/r101_b256_gen1_fallback.log 2>&1
for f in r101_b256 r101_b256_gen1_fallback; do echo $f; tail -1 gpurun_out/bo/$f.log | cut -c1-420; done
