#!/bin/bash
# X65 os8 B64 v33 kernel breakdown
set -o pipefail
mkdir -p gpurun_out/ar
export TMPDIR=/tmp
(cd /tmp && timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/ar/prof_x65 -- python $GRAFT_REPO_ROOT/bench.py --model deeplabv3plus_xception65 --output-stride 8 --batch 64 --steps 4 --warmup 1 > $GRAFT_REPO_ROOT/gpurun_out/ar/prof_x65.log 2>&1) &&
python tools/summarize_prof.py $(find gpurun_out/ar/prof_x65 -name "*kernel_stats.csv" | head -1) 5 "Xception-65 OS8 B64, v33 (1 warmup + 4 steps)" > gpurun_out/ar/x65_b64_profile.md 2>&1
rm -rf gpurun_out/ar/prof_x65
head -46 gpurun_out/ar/x65_b64_profile.md
