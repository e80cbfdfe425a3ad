#!/bin/bash
# final tree: the small-batch graph-step rows and DANet B32 once more
set -u
O=gpurun_out/r7u; mkdir -p $O
run() { name=$1; shift; timeout -k 10 300 python bench.py --steps 30 --warmup 6 "$@" > $O/$name.log 2>&1 || { tail -5 $O/$name.log; exit 1; }; echo "$name $(grep -o '"value": [0-9.]*' $O/$name.log)"; }
run r101_b16_eager --batch 16
run r101_b16_graph --batch 16 --graph-step
run r101_b8_graph --batch 8 --graph-step
run danet_b4_graph --task instance --model danet_resnet101 --crop 512 --batch 4 --graph-step
run danet_b32 --task instance --model danet_resnet101 --crop 512 --batch 32
