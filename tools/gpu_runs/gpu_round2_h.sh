#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/h
export TMPDIR=/tmp DPAMD_HEARTBEAT_S=45
timeout -k 10 400 python bench.py --model deeplabv3plus_xception65 --output-stride 8 --batch 64 --steps 8 --warmup 3 --phases 2>&1 | tee gpurun_out/h/x65_b64.log | tail -3 || exit 1
cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/h/prof -- python $GRAFT_REPO_ROOT/bench.py --model deeplabv3plus_xception65 --output-stride 8 --batch 64 --steps 4 --warmup 1 > $GRAFT_REPO_ROOT/gpurun_out/h/prof.log 2>&1
cd $GRAFT_REPO_ROOT && python tools/summarize_prof.py $(find gpurun_out/h/prof -name "*kernel_stats.csv" | head -1) 5 "Xception-65 OS8 B64 (1 warmup + 4 steps, tile choices preloaded)" > gpurun_out/h/x65_b64_profile.md 2>&1; head -48 gpurun_out/h/x65_b64_profile.md
