#!/bin/bash
# phase stamps of a conv block's life (tools/stamp_conv.py; diagnostic library built beforehand on the CPU box)
set -o pipefail
mkdir -p gpurun_out/ab
timeout -k 10 200 python -u tools/stamp_conv.py --batch 64 --cfgs 5,9,3 728:728:65:1 1536:1536:65:1 > gpurun_out/ab/stamp_x65.log 2>&1 &&
timeout -k 10 200 python -u tools/stamp_conv.py --batch 128 --cfgs 16,9,5 256:1024:33:1 1024:256:33:1 256:256:33:3 > gpurun_out/ab/stamp_r101.log 2>&1
cat gpurun_out/ab/stamp_x65.log gpurun_out/ab/stamp_r101.log
