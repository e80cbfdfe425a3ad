#!/bin/bash
# store-loop batching A/B with phase stamps: residual loop (batched loads) and BN-backward store loop at batch depth 4 vs 8
set -o pipefail
mkdir -p gpurun_out/ad
L8=distributedpytorch_amd/csrc/build/libdpamd_stamp_g8.so
timeout -k 10 200 python -u tools/stamp_conv.py --batch 64 --cfgs 5,3 --modes res,resaff,bn 728:728:65:1 > gpurun_out/ad/x65_g4.log 2>&1 &&
timeout -k 10 200 python -u tools/stamp_conv.py --lib $L8 --batch 64 --cfgs 5,3 --modes bn 728:728:65:1 > gpurun_out/ad/x65_g8.log 2>&1 &&
timeout -k 10 200 python -u tools/stamp_conv.py --batch 128 --cfgs 16,9 --modes res,bn 256:1024:33:1 1024:256:33:1 64:256:129:1 > gpurun_out/ad/r101_g4.log 2>&1 &&
timeout -k 10 200 python -u tools/stamp_conv.py --lib $L8 --batch 128 --cfgs 16,9 --modes bn 256:1024:33:1 1024:256:33:1 64:256:129:1 > gpurun_out/ad/r101_g8.log 2>&1
tail -n +2 gpurun_out/ad/x65_g4.log gpurun_out/ad/x65_g8.log gpurun_out/ad/r101_g4.log gpurun_out/ad/r101_g8.log
