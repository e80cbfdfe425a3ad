#!/bin/bash
# v31 (batched residual store loop, BN store loop 8 rows deep): GPU tests, stagger re-test on the HBM-bound store loops, benches
set -o pipefail
mkdir -p gpurun_out/ae
timeout -k 10 600 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread > gpurun_out/ae/tests.log 2>&1 || { tail -30 gpurun_out/ae/tests.log; exit 1; }
tail -3 gpurun_out/ae/tests.log
timeout -k 10 200 python -u tools/stamp_conv.py --batch 64 --cfgs 5 --modes resaff,bn --stagger "0,0;0,6;0,12;0,20" 728:728:65:1 > gpurun_out/ae/stagger_x65.log 2>&1 &&
timeout -k 10 300 python bench.py --batch 128 --steps 20 --warmup 5 > gpurun_out/ae/r101_b128.log 2>&1 &&
timeout -k 10 300 python bench.py --model deeplabv3plus_xception65 --output-stride 8 --batch 64 --steps 10 --warmup 3 > gpurun_out/ae/x65_b64.log 2>&1
tail -n +2 gpurun_out/ae/stagger_x65.log; tail -1 gpurun_out/ae/r101_b128.log; tail -1 gpurun_out/ae/x65_b64.log
