#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/aa
export TMPDIR=/tmp DPAMD_HEARTBEAT_S=45
cd /tmp && timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/aa/prof -- python $GRAFT_REPO_ROOT/bench.py --batch 128 --steps 4 --warmup 1 > $GRAFT_REPO_ROOT/gpurun_out/aa/prof.log 2>&1
cd $GRAFT_REPO_ROOT && python tools/summarize_prof.py $(find gpurun_out/aa/prof -name "*kernel_stats.csv" | head -1) 5 "DeepLabv3+ R101 513x513 B128, v30 (1 warmup + 4 steps)" > gpurun_out/aa/r101_b128_profile.md 2>&1; head -50 gpurun_out/aa/r101_b128_profile.md
