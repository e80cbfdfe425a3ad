#!/bin/bash
# v32: positivity bit mask for residual blocks (BN apply writes it, fused dgrad store loop reads it): tests + benches, A/B
set -o pipefail
mkdir -p gpurun_out/ah
timeout -k 10 600 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread > gpurun_out/ah/tests.log 2>&1 || { tail -30 gpurun_out/ah/tests.log; exit 1; }
tail -2 gpurun_out/ah/tests.log
timeout -k 10 300 python bench.py --batch 128 --steps 20 --warmup 5 > gpurun_out/ah/r101_b128_bits.log 2>&1 &&
DPAMD_BN_MASK_BITS=0 timeout -k 10 300 python bench.py --batch 128 --steps 20 --warmup 5 > gpurun_out/ah/r101_b128_nobits.log 2>&1 &&
timeout -k 10 300 python bench.py --batch 128 --steps 20 --warmup 5 > gpurun_out/ah/r101_b128_bits2.log 2>&1
for f in bits nobits bits2; do tail -1 gpurun_out/ah/r101_b128_$f.log | cut -c1-200; done
