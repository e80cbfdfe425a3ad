#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/l
timeout -k 10 400 python -u -m pytest tests/test_gpu_kernels.py tests/test_gpu_sepconv.py -x -q --timeout 120 --timeout-method thread 2>&1 | tee gpurun_out/l/tests.log | tail -12 &&
DPAMD_HEARTBEAT_S=30 timeout -k 10 500 python -u bench.py --model deeplabv3plus_xception65 --output-stride 8 --batch 64 --steps 10 --warmup 4 --phases 2>&1 | tee gpurun_out/l/x65_b64.log | tail -4
