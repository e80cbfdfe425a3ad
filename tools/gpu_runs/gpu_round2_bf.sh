#!/bin/bash
# rolling-window depthwise kernels without compiler-inserted vmcnt(0) before LDS reads (inline-asm ds_read, counted waits,
# branch-free buffer loads/stores): tests, micro-benchmark, X65 step
set -o pipefail
mkdir -p gpurun_out/bf
export TMPDIR=/tmp DPAMD_HEARTBEAT_S=45
timeout -k 10 400 python -u -m pytest tests/test_gpu_kernels.py tests/test_gpu_sepconv.py tests/test_gpu_hardening.py -m gpu -x -q -k "depthwise or xception or sepconv or folded" --timeout 120 --timeout-method thread > gpurun_out/bf/tests.log 2>&1 || { tail -40 gpurun_out/bf/tests.log; exit 1; }
tail -2 gpurun_out/bf/tests.log
X="python bench.py --model deeplabv3plus_xception65 --output-stride 8 --batch 64 --steps 10 --warmup 3"
timeout -k 10 200 python -u tools/bench_dw.py --batch 64 --gen 4 > gpurun_out/bf/dw_gen4.log 2>&1 &&
timeout -k 10 300 $X > gpurun_out/bf/x65_a.log 2>&1 &&
timeout -k 10 300 $X > gpurun_out/bf/x65_b.log 2>&1
grep -v amdgpu gpurun_out/bf/dw_gen4.log
for f in a b; do tail -1 gpurun_out/bf/x65_$f.log | cut -c1-200; done
