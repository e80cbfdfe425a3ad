#!/bin/bash
# depthwise micro-benchmark (launchers timed directly): generation 4 vs 5
set -o pipefail
mkdir -p gpurun_out/ak
DPAMD_DWCONV_GEN=4 timeout -k 10 200 python -u tools/bench_dw.py --batch 64 > gpurun_out/ak/dw_gen4.log 2>&1 &&
timeout -k 10 200 python -u tools/bench_dw.py --batch 64 > gpurun_out/ak/dw_gen5.log 2>&1
grep -v amdgpu gpurun_out/ak/dw_gen4.log; grep -v amdgpu gpurun_out/ak/dw_gen5.log
