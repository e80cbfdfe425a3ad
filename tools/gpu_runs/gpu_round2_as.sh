#!/bin/bash
# column bands in the rolling-window depthwise kernels (wide entry-flow maps): tests, X65 step
set -o pipefail
mkdir -p gpurun_out/as
timeout -k 10 300 python -u -m pytest tests/test_gpu_kernels.py tests/test_gpu_sepconv.py tests/test_gpu_hardening.py -m gpu -x -q -k "depthwise or xception or sepconv or folded" --timeout 120 --timeout-method thread > gpurun_out/as/tests.log 2>&1 || { tail -40 gpurun_out/as/tests.log; exit 1; }
tail -2 gpurun_out/as/tests.log
X="python bench.py --model deeplabv3plus_xception65 --output-stride 8 --batch 64 --steps 10 --warmup 3"
timeout -k 10 300 $X > gpurun_out/as/x65_a.log 2>&1 && timeout -k 10 300 $X > gpurun_out/as/x65_b.log 2>&1
for f in a b; do tail -1 gpurun_out/as/x65_$f.log | cut -c1-200; done
