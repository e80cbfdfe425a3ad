#!/bin/bash
# predict.py --gpus N rehearsed with ranks sharing the one GPU (gloo rendezvous; no collectives on device tensors)
set -u
O=gpurun_out/r7k; mkdir -p $O
timeout -k 10 300 python -u predict.py --synthetic 128 --batch 8 --no-write > $O/one_rank.log 2>&1 || { tail -5 $O/one_rank.log; exit 1; }
tail -1 $O/one_rank.log | cut -c1-260
for n in 2 4; do
  DPAMD_DIST_BACKEND=gloo timeout -k 10 300 python -u predict.py --synthetic 128 --batch 8 --gpus $n --output $O/out_$n > $O/ranks_$n.log 2>&1 || { grep -v Warning $O/ranks_$n.log | tail -12 | cut -c1-250; exit 1; }
  grep '^{' $O/ranks_$n.log | cut -c1-260; ls $O/out_$n | wc -l; rm -rf $O/out_$n
done
DPAMD_DIST_BACKEND=gloo timeout -k 10 300 python -u predict.py --task instance --synthetic 64 --batch 8 --gpus 2 --no-write > $O/instance_ranks_2.log 2>&1 || { grep -v Warning $O/instance_ranks_2.log | tail -12 | cut -c1-250; exit 1; }
grep '^{' $O/instance_ranks_2.log | cut -c1-260
