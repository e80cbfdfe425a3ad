#!/bin/bash
# experiment: Xception-65 middle flow at 768 physical channels (cache-line-aligned rows) against 728, same box
set -o pipefail
mkdir -p gpurun_out/ba
export TMPDIR=/tmp DPAMD_HEARTBEAT_S=45
timeout -k 10 300 python bench.py --model deeplabv3plus_xception65 --output-stride 8 --batch 64 --steps 10 --warmup 3 > gpurun_out/ba/x65_728.log 2>&1 &&
DPAMD_X65_MID=768 timeout -k 10 400 python bench.py --model deeplabv3plus_xception65 --output-stride 8 --batch 64 --steps 10 --warmup 4 > gpurun_out/ba/x65_768.log 2>&1 &&
(cd /tmp && DPAMD_X65_MID=768 timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/ba/prof_x65 -- python $GRAFT_REPO_ROOT/bench.py --model deeplabv3plus_xception65 --output-stride 8 --batch 64 --steps 4 --warmup 2 > $GRAFT_REPO_ROOT/gpurun_out/ba/prof_x65.log 2>&1) &&
python tools/summarize_prof.py $(find gpurun_out/ba/prof_x65 -name "*kernel_stats.csv" | head -1) 6 "Xception-65 OS8 B64, middle flow 768 wide (2 warmup + 4 steps)" > gpurun_out/ba/x65_768_profile.md 2>&1
rm -rf gpurun_out/ba/prof_x65
for f in x65_728 x65_768; do tail -1 gpurun_out/ba/$f.log | cut -c1-210; done
head -40 gpurun_out/ba/x65_768_profile.md
