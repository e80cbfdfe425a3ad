#!/bin/bash
# DANet-R101 512 instance task at per-GPU batch 64 / 96
set -o pipefail
mkdir -p gpurun_out/bt
export TMPDIR=/tmp DPAMD_HEARTBEAT_S=45
for b in 64 96; do
DPAMD_CONV_TUNE_FILE=$GRAFT_REPO_ROOT/gpurun_out/bt/danet_b${b}_tune.json timeout -k 10 500 python bench.py --task instance --model danet_resnet101 --crop 512 --in-channels 4 --batch $b --steps 8 --warmup 3 > gpurun_out/bt/danet_b$b.log 2>&1; tail -1 gpurun_out/bt/danet_b$b.log | cut -c1-220
done
