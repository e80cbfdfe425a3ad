#!/bin/bash
# depthwise batch chunks: tests, Xception-65 OS8 at per-GPU batch 128 (depthwise tensors past 2 GiB) and 64
set -o pipefail
mkdir -p gpurun_out/cb
export TMPDIR=/tmp DPAMD_HEARTBEAT_S=45
timeout -k 10 500 python -u -m pytest tests/test_gpu_hardening.py tests/test_gpu_sepconv.py tests/test_gpu_kernels.py -m gpu -x -q -k "chunk or depthwise or sepconv or xception" --timeout 180 --timeout-method thread > gpurun_out/cb/tests.log 2>&1 || { tail -40 gpurun_out/cb/tests.log; exit 1; }
tail -1 gpurun_out/cb/tests.log
DPAMD_CONV_TUNE_FILE=$GRAFT_REPO_ROOT/gpurun_out/cb/x65_b128_tune.json timeout -k 10 500 python bench.py --model deeplabv3plus_xception65 --output-stride 8 --batch 128 --steps 8 --warmup 3 > gpurun_out/cb/x65_b128.log 2>&1; tail -1 gpurun_out/cb/x65_b128.log | cut -c1-220
timeout -k 10 300 python bench.py --model deeplabv3plus_xception65 --output-stride 8 --batch 64 --steps 10 --warmup 3 > gpurun_out/cb/x65_b64.log 2>&1; tail -1 gpurun_out/cb/x65_b64.log | cut -c1-220
