#!/bin/bash
# two-rank rehearsals of the GPU data-parallel path on ONE GPU (gloo between processes, DPAMD_DEBUG cross-rank checks) on v38:
# R101 with SyncBN over the one-shot p2p kernel, Xception-65 (768-wide stored flow) plain DDP
mkdir -p gpurun_out
bash tools/rehearse_gloo.sh v38_r101_syncbn DPAMD_DEBUG=1 DPAMD_SYNCBN_P2P=1 -- --batch 8 --crop 257 --steps 4 --warmup 2 --sync-bn
bash tools/rehearse_gloo.sh v38_x65 DPAMD_DEBUG=1 -- --model deeplabv3plus_xception65 --output-stride 8 --batch 4 --crop 257 --steps 4 --warmup 2
