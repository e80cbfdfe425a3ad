#!/bin/bash
# position-attention backward with prefetched lse/delta and pipelined fragment reads: tests + DANet bench
set -o pipefail
mkdir -p gpurun_out/ay
timeout -k 10 300 python -u -m pytest tests/test_gpu_attention.py -m gpu -x -q --timeout 120 --timeout-method thread > gpurun_out/ay/tests.log 2>&1 || { tail -40 gpurun_out/ay/tests.log; exit 1; }
tail -2 gpurun_out/ay/tests.log
D="python bench.py --task instance --model danet_resnet101 --crop 512 --in-channels 4 --steps 10 --warmup 3"
timeout -k 10 300 $D --batch 16 > gpurun_out/ay/danet_b16.log 2>&1 && timeout -k 10 300 $D --batch 32 > gpurun_out/ay/danet_b32.log 2>&1
for f in b16 b32; do tail -1 gpurun_out/ay/danet_$f.log | cut -c1-200; done
