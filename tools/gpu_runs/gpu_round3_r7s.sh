#!/bin/bash
# convergence parity re-run on the round-3 tree (BN3 fold, dropout in the dgrad prologue, direct depthwise sinks, keep-alive
# sections): native only, texture task, same settings and seeds as profiles/r2_parity_texture_summary.md; then the same R101
# run twice under DPAMD_DETERMINISTIC=1 (same seed): the two must agree to the last digit
set -o pipefail
mkdir -p gpurun_out/r7s
export TMPDIR=/tmp
X="--task texture --model deeplabv3plus_xception65 --output-stride 8 --crop 257 --batch 16 --iters 700 --train-images 192 --val-images 96 --impls native"
R="--task texture --model deeplabv3plus_resnet101 --crop 257 --batch 16 --iters 300 --train-images 192 --val-images 48 --impls native"
timeout -k 10 400 python -u tools/parity_run.py $R --seeds 0,1,2,3,4 2>gpurun_out/r7s/r101.err | tee gpurun_out/r7s/r101_native_v57_seeds0-4.log | cut -c1-100 | tail -2 &&
timeout -k 10 600 python -u tools/parity_run.py $X --seeds 0,1,2,3,4,5 2>gpurun_out/r7s/x65.err | tee gpurun_out/r7s/x65_native_v57_seeds0-5.log | cut -c1-100 | tail -2 &&
DPAMD_DETERMINISTIC=1 timeout -k 10 300 python -u tools/parity_run.py $R --seeds 0,0 2>gpurun_out/r7s/r101_det.err | tee gpurun_out/r7s/r101_native_v57_deterministic_twice.log | cut -c1-100 | tail -2
rc=$?
for f in gpurun_out/r7s/*.log; do tail -1 $f | cut -c1-300; done
exit $rc
