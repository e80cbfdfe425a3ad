#!/bin/bash
# weight gradient on v_mfma_f32_16x16x32_bf16 (DPAMD_WGRAD_MFMA=16) vs 32x32x16: tests, phase stamps, R101 step
set -o pipefail
mkdir -p gpurun_out/au
DPAMD_WGRAD_MFMA=16 timeout -k 10 300 python -u -m pytest tests/test_gpu_kernels.py tests/test_gpu_hardening.py -m gpu -x -q -k "wgrad or headline or conv2d_module or training_step" --timeout 120 --timeout-method thread > gpurun_out/au/tests16.log 2>&1 || { tail -40 gpurun_out/au/tests16.log; exit 1; }
tail -2 gpurun_out/au/tests16.log
S="256:1024:33:1 1024:256:33:1 256:256:33:3 512:2048:33:1 304:256:129:3 2048:256:33:3:12"
timeout -k 10 250 python -u tools/stamp_conv.py --wgrad --batch 128 $S > gpurun_out/au/wgrad32.log 2>&1 &&
DPAMD_WGRAD_MFMA=16 timeout -k 10 250 python -u tools/stamp_conv.py --wgrad --batch 128 $S > gpurun_out/au/wgrad16.log 2>&1 &&
timeout -k 10 300 python bench.py --batch 128 --steps 20 --warmup 5 > gpurun_out/au/r101_mfma32.log 2>&1 &&
DPAMD_WGRAD_MFMA=16 timeout -k 10 300 python bench.py --batch 128 --steps 20 --warmup 5 > gpurun_out/au/r101_mfma16.log 2>&1 &&
timeout -k 10 300 python bench.py --batch 128 --steps 20 --warmup 5 > gpurun_out/au/r101_mfma32b.log 2>&1 &&
DPAMD_WGRAD_MFMA=16 timeout -k 10 300 python bench.py --batch 128 --steps 20 --warmup 5 > gpurun_out/au/r101_mfma16b.log 2>&1
grep -v amdgpu gpurun_out/au/wgrad32.log; grep -v amdgpu gpurun_out/au/wgrad16.log
for f in mfma32 mfma16 mfma32b mfma16b; do echo $f; tail -1 gpurun_out/au/r101_$f.log | cut -c1-180; done
