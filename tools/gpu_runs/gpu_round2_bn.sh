#!/bin/bash
# synchronised BatchNorm's launch choreography on ONE GPU (1-rank RCCL group, collectives forced): overhead per step at B128
set -o pipefail
mkdir -p gpurun_out/bn
export TMPDIR=/tmp DPAMD_HEARTBEAT_S=45
R="python bench.py --batch 128 --steps 20 --warmup 5"
timeout -k 10 300 $R > gpurun_out/bn/plain.log 2>&1 &&
DPAMD_DDP_FORCE=1 timeout -k 10 300 $R > gpurun_out/bn/ddp_forced.log 2>&1 &&
DPAMD_DDP_FORCE=1 DPAMD_SYNCBN_FORCE=1 timeout -k 10 300 $R --sync-bn > gpurun_out/bn/syncbn_forced_rccl.log 2>&1
for f in plain ddp_forced syncbn_forced_rccl; do echo $f; tail -1 gpurun_out/bn/$f.log | cut -c1-1200; done
