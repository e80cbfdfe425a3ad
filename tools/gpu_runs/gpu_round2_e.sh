#!/bin/bash
# folded sepconv: numerics, then Xception-65 bench at B64
set -o pipefail
mkdir -p gpurun_out/e
timeout -k 10 400 python -u -m pytest tests/test_gpu_sepconv.py -x -v --timeout 120 --timeout-method thread 2>&1 | tee gpurun_out/e/sepconv_tests.log | tail -25 &&
DPAMD_HEARTBEAT_S=30 timeout -k 10 500 python -u bench.py --model deeplabv3plus_xception65 --output-stride 8 --batch 64 --steps 10 --warmup 4 2>&1 | tee gpurun_out/e/x65_b64_fold.log | tail -5
