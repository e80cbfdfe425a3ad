#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/u
timeout -k 10 1100 python -u tools/parity_run.py --task texture --model deeplabv3plus_resnet101 --crop 257 --batch 16 --iters 300 --train-images 192 --val-images 96 --seeds 0,1,2,3,4 2>gpurun_out/u/r101.err | tee gpurun_out/u/parity_texture_r101.log | cut -c1-150 | tail -12
