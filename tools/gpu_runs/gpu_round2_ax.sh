#!/bin/bash
# X65 os8: per-GPU batch 96 vs 64
set -o pipefail
mkdir -p gpurun_out/ax
timeout -k 10 500 python bench.py --model deeplabv3plus_xception65 --output-stride 8 --batch 96 --steps 8 --warmup 3 --phases > gpurun_out/ax/x65_b96.log 2>&1
tail -3 gpurun_out/ax/x65_b96.log | cut -c1-400
