#!/bin/bash
# convergence parity re-run on v38 (Xception-65 flow stored 768 wide, depthwise inline-asm reads, weight-gradient DMA spread,
# store loops with unconditional loads): native only, texture task, same settings as profiles/r2_parity_texture_summary.md
set -o pipefail
mkdir -p gpurun_out/cf
export TMPDIR=/tmp
X="--task texture --model deeplabv3plus_xception65 --output-stride 8 --crop 257 --batch 16 --iters 700 --train-images 192 --val-images 96 --impls native"
R="--task texture --model deeplabv3plus_resnet101 --crop 257 --batch 16 --iters 300 --train-images 192 --val-images 48 --impls native"
timeout -k 10 600 python -u tools/parity_run.py $X --seeds 0,1,2,3,4,5 2>gpurun_out/cf/x65.err | tee gpurun_out/cf/x65_native_v38_seeds0-5.log | cut -c1-100 | tail -2 &&
timeout -k 10 400 python -u tools/parity_run.py $R --seeds 0,1,2,3,4 2>gpurun_out/cf/r101.err | tee gpurun_out/cf/r101_native_v38_seeds0-4.log | cut -c1-100 | tail -2
tail -1 gpurun_out/cf/x65_native_v38_seeds0-5.log | cut -c1-300; tail -1 gpurun_out/cf/r101_native_v38_seeds0-4.log | cut -c1-300
