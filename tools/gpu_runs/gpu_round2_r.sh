#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r
timeout -k 10 1000 python -u tools/parity_run.py --task texture --model deeplabv3plus_resnet101 --crop 257 --batch 16 --iters 400 --train-images 192 --val-images 48 2>&1 | tee gpurun_out/r/parity_texture_r101.log | tail -5
