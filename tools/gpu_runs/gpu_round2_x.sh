#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/x
timeout -k 10 1150 python -u tools/parity_run.py --task texture --model deeplabv3plus_xception65 --output-stride 8 --crop 257 --batch 16 --iters 700 --train-images 192 --val-images 96 --seeds 0,1,2 --impls stock_fp32 2>gpurun_out/x/x65.err | tee gpurun_out/x/parity_texture_x65_700_fp32.log | cut -c1-150 | tail -8
