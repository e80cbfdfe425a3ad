#!/bin/bash
# Xception-65 (728-channel flow stored 768 wide, packaged tile table): bench, unpadded bench on the same box, profile
set -o pipefail
mkdir -p gpurun_out/bc
export TMPDIR=/tmp DPAMD_HEARTBEAT_S=45
timeout -k 10 300 python bench.py --model deeplabv3plus_xception65 --output-stride 8 --batch 64 --steps 10 --warmup 3 > gpurun_out/bc/x65_768.log 2>&1 &&
DPAMD_X65_PAD=0 timeout -k 10 300 python bench.py --model deeplabv3plus_xception65 --output-stride 8 --batch 64 --steps 10 --warmup 3 > gpurun_out/bc/x65_728.log 2>&1 &&
timeout -k 10 300 python bench.py --model deeplabv3plus_xception65 --output-stride 8 --batch 64 --steps 10 --warmup 3 > gpurun_out/bc/x65_768_b.log 2>&1 &&
(cd /tmp && timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/bc/prof_x65 -- python $GRAFT_REPO_ROOT/bench.py --model deeplabv3plus_xception65 --output-stride 8 --batch 64 --steps 4 --warmup 1 > $GRAFT_REPO_ROOT/gpurun_out/bc/prof_x65.log 2>&1) &&
python tools/summarize_prof.py $(find gpurun_out/bc/prof_x65 -name "*kernel_stats.csv" | head -1) 5 "Xception-65 OS8 B64, v36: 728-channel flow stored 768 wide (1 warmup + 4 steps)" > gpurun_out/bc/x65_b64_profile.md 2>&1
rm -rf gpurun_out/bc/prof_x65
for f in x65_768 x65_728 x65_768_b; do tail -1 gpurun_out/bc/$f.log | cut -c1-210; done
head -32 gpurun_out/bc/x65_b64_profile.md
