#!/bin/bash
# depthwise forward / data gradient on the matrix cores (generation 6): tests, micro-benchmark vs generation 4
set -o pipefail
mkdir -p gpurun_out/ao
timeout -k 10 300 python -u -m pytest tests/test_gpu_kernels.py tests/test_gpu_sepconv.py -m gpu -x -q -k "depthwise or xception" --timeout 120 --timeout-method thread > gpurun_out/ao/tests.log 2>&1 || { tail -40 gpurun_out/ao/tests.log; exit 1; }
tail -2 gpurun_out/ao/tests.log
timeout -k 10 200 python -u tools/bench_dw.py --batch 64 --gen 4 > gpurun_out/ao/dw_gen4.log 2>&1 &&
timeout -k 10 200 python -u tools/bench_dw.py --batch 64 --gen 6 > gpurun_out/ao/dw_gen6.log 2>&1
grep -v amdgpu gpurun_out/ao/dw_gen4.log; grep -v amdgpu gpurun_out/ao/dw_gen6.log
