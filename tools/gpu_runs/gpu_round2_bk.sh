#!/bin/bash
# is the weight-gradient K loop bound by where its operands come from?  per-K-step time for L2-resident, Infinity-Cache-resident
# and HBM-sized operands of the same 3x3 256->256 and 1x1 1024->256 problems (batch 128; map edge 12 / 33 / 66)
set -o pipefail
mkdir -p gpurun_out/bk
timeout -k 10 300 python -u tools/stamp_conv.py --wgrad --batch 128 256:256:12:3 256:256:33:3 256:256:66:3 1024:256:12:1 1024:256:33:1 1024:256:66:1 > gpurun_out/bk/wgrad_residency.txt 2>&1
grep -v amdgpu gpurun_out/bk/wgrad_residency.txt
