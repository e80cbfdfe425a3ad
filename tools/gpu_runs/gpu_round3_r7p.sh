#!/bin/bash
# the driver's launch forms on the final tree: torchrun with one rank over RCCL, and two / four ranks sharing the GPU over gloo
set -u
O=gpurun_out/r7p; mkdir -p $O
timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 1 --steps 10 --warmup 3 > $O/torchrun_1.log 2>&1 || { tail -8 $O/torchrun_1.log | cut -c1-250; exit 1; }
grep -o '"value": [0-9.]*\|"backend": "[a-z]*"\|"n_gpus": [0-9]*' $O/torchrun_1.log | tr '\n' ' '; echo
for n in 2 4; do
DPAMD_DIST_BACKEND=gloo timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port 2962$n bench.py --gpus $n --steps 4 --warmup 2 --batch 32 > $O/torchrun_gloo_$n.log 2>&1 || { grep -v Warning $O/torchrun_gloo_$n.log | tail -12 | cut -c1-250; exit 1; }
grep -o '"value": [0-9.]*\|"backend": "[a-z]*"\|"n_gpus": [0-9]*\|"exposed_after_backward_ms_max_over_ranks": [0-9.]*' $O/torchrun_gloo_$n.log | tr '\n' ' '; echo
done
DPAMD_DIST_BACKEND=gloo timeout -k 10 400 python bench.py --gpus 2 --steps 4 --warmup 2 --batch 32 --sync-bn > $O/selflaunch_gloo_2_syncbn.log 2>&1 || { grep -v Warning $O/selflaunch_gloo_2_syncbn.log | tail -12 | cut -c1-250; exit 1; }
grep -o '"value": [0-9.]*\|"backend": "[a-z]*"\|"n_gpus": [0-9]*\|"sync_bn": [a-z]*' $O/selflaunch_gloo_2_syncbn.log | tr '\n' ' '; echo
