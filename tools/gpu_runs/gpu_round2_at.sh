#!/bin/bash
# phase stamps of the weight-gradient blocks (R101 B128 layers, X65 pointwise B64)
set -o pipefail
mkdir -p gpurun_out/at
timeout -k 10 250 python -u tools/stamp_conv.py --wgrad --batch 128 256:1024:33:1 1024:256:33:1 256:256:33:3 64:256:129:1 256:64:129:1 64:64:129:3 512:2048:33:1 304:256:129:3 2048:256:33:3:12 > gpurun_out/at/wgrad_r101.log 2>&1 &&
timeout -k 10 250 python -u tools/stamp_conv.py --wgrad --batch 64 728:728:65:1 > gpurun_out/at/wgrad_x65.log 2>&1
grep -v amdgpu gpurun_out/at/wgrad_r101.log; grep -v amdgpu gpurun_out/at/wgrad_x65.log
