#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/z
A="--task texture --model deeplabv3plus_xception65 --output-stride 8 --crop 257 --batch 16 --iters 700 --train-images 192 --val-images 96"
timeout -k 10 400 python -u tools/parity_run.py $A --impls native --seeds 3,4,5,6,7,8 2>gpurun_out/z/native.err | tee gpurun_out/z/x65_native_seeds3-8.log | cut -c1-120 | tail -3 &&
timeout -k 10 750 python -u tools/parity_run.py $A --impls stock --seeds 3,4,5 2>gpurun_out/z/stock.err | tee gpurun_out/z/x65_stock_seeds3-5.log | cut -c1-120 | tail -3
