#!/bin/bash
# depthwise weight gradient on the matrix cores (dwconv5): tests, micro-benchmark gen 4 vs 5, X65 bench
set -o pipefail
mkdir -p gpurun_out/aj
timeout -k 10 300 python -u -m pytest tests/test_gpu_kernels.py tests/test_gpu_sepconv.py -m gpu -x -q -k "depthwise or sepconv or xception" --timeout 120 --timeout-method thread > gpurun_out/aj/tests.log 2>&1 || { tail -40 gpurun_out/aj/tests.log; exit 1; }
tail -2 gpurun_out/aj/tests.log
DPAMD_DWCONV_GEN=4 timeout -k 10 200 python -u tools/bench_dw.py --batch 64 > gpurun_out/aj/dw_gen4.log 2>&1 &&
timeout -k 10 200 python -u tools/bench_dw.py --batch 64 > gpurun_out/aj/dw_gen5.log 2>&1 &&
timeout -k 10 300 python bench.py --model deeplabv3plus_xception65 --output-stride 8 --batch 64 --steps 10 --warmup 3 > gpurun_out/aj/x65_b64.log 2>&1
grep -v amdgpu gpurun_out/aj/dw_gen4.log; grep -v amdgpu gpurun_out/aj/dw_gen5.log; tail -1 gpurun_out/aj/x65_b64.log | cut -c1-200
