#!/bin/bash
# what is slow in Xception-65 OS8 at per-GPU batch 128?
set -o pipefail
mkdir -p gpurun_out/cc
export TMPDIR=/tmp DPAMD_HEARTBEAT_S=45
(cd /tmp && timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/cc/prof -- python $GRAFT_REPO_ROOT/bench.py --model deeplabv3plus_xception65 --output-stride 8 --batch 128 --steps 3 --warmup 2 > $GRAFT_REPO_ROOT/gpurun_out/cc/prof.log 2>&1) &&
python tools/summarize_prof.py $(find gpurun_out/cc/prof -name "*kernel_stats.csv" | head -1) 5 "X65 OS8 B128 (2 warmup incl. autotune + 3 steps)" > gpurun_out/cc/profile.md 2>&1
rm -rf gpurun_out/cc/prof
head -36 gpurun_out/cc/profile.md | cut -c1-150
