#!/bin/bash
# Multi-rank rehearsal of the GPU data-parallel path on ONE GPU: ranks share the device and talk over
# gloo (RCCL refuses two ranks on a device).  usage: tools/rehearse_gloo.sh <tag> [ENV=VAL ...] -- <bench args>
tag=$1; shift
envs=()
while [ "$1" != "--" ] && [ $# -gt 0 ]; do envs+=("$1"); shift; done
shift
out=gpurun_out/rehearse_${tag}.log
( while true; do sleep 25; echo "hb $(date +%T)"; done ) &
hb=$!
env DPAMD_HANG_DUMP_S=150 DPAMD_DIST_BACKEND=gloo "${envs[@]}" timeout -k 5 170 python -m torch.distributed.run --nnodes=1 \
    --nproc-per-node ${NPROC:-2} --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus ${NPROC:-2} "$@" > $out 2>&1
rc=$?
kill $hb
echo "== $tag rc=$rc"
grep -E '"value"|Timeout|File ".*distributedpytorch_amd|File ".*bench.py' $out | cut -c1-160 | head -12
exit 0
