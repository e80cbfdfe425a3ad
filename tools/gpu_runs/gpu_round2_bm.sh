#!/bin/bash
# DANet-R101 512 instance task: three autotune runs per batch (16, 32) for the packaged tile table
set -o pipefail
mkdir -p gpurun_out/bm
export TMPDIR=/tmp DPAMD_HEARTBEAT_S=45
for b in 16 32; do for i in 1 2 3; do
  DPAMD_CONV_TUNE_FILE=$GRAFT_REPO_ROOT/gpurun_out/bm/danet_b${b}_tune_$i.json timeout -k 10 400 python bench.py --task instance --model danet_resnet101 --crop 512 --in-channels 4 --batch $b --steps 8 --warmup 3 > gpurun_out/bm/danet_b${b}_$i.log 2>&1 || { tail -20 gpurun_out/bm/danet_b${b}_$i.log; exit 1; }
  tail -1 gpurun_out/bm/danet_b${b}_$i.log | cut -c1-200
done; done
