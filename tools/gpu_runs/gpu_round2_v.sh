#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/v
timeout -k 10 1150 python -u tools/parity_run.py --task texture --model deeplabv3plus_resnet101 --crop 257 --batch 16 --iters 900 --train-images 192 --val-images 96 --seeds 0,1,2 2>gpurun_out/v/r101.err | tee gpurun_out/v/parity_texture_r101_900.log | cut -c1-150 | tail -8
