#!/bin/bash
# depthwise generations 4 / 6 with the XCD-aware band-major block order (ragged channel counts only): tests, micro-benchmark, X65 step
set -o pipefail
mkdir -p gpurun_out/ap
timeout -k 10 300 python -u -m pytest tests/test_gpu_kernels.py tests/test_gpu_sepconv.py -m gpu -x -q -k "depthwise or xception or sepconv or folded" --timeout 120 --timeout-method thread > gpurun_out/ap/tests.log 2>&1 || { tail -40 gpurun_out/ap/tests.log; exit 1; }
tail -2 gpurun_out/ap/tests.log
X="python bench.py --model deeplabv3plus_xception65 --output-stride 8 --batch 64 --steps 10 --warmup 3"
timeout -k 10 200 python -u tools/bench_dw.py --batch 64 --gen 4 > gpurun_out/ap/dw_gen4.log 2>&1 &&
timeout -k 10 200 python -u tools/bench_dw.py --batch 64 --gen 6 > gpurun_out/ap/dw_gen6.log 2>&1 &&
DPAMD_DWCONV_GEN=4 timeout -k 10 300 $X > gpurun_out/ap/x65_gen4.log 2>&1 &&
DPAMD_DWCONV_GEN=6 timeout -k 10 300 $X > gpurun_out/ap/x65_gen6.log 2>&1 &&
DPAMD_DWCONV_GEN=4 timeout -k 10 300 $X > gpurun_out/ap/x65_gen4b.log 2>&1 &&
DPAMD_DWCONV_GEN=6 timeout -k 10 300 $X > gpurun_out/ap/x65_gen6b.log 2>&1
grep -v amdgpu gpurun_out/ap/dw_gen4.log; grep -v amdgpu gpurun_out/ap/dw_gen6.log
for f in gen4 gen6 gen4b gen6b; do echo $f; tail -1 gpurun_out/ap/x65_$f.log | cut -c1-200; done
