#!/bin/bash
# DANet-R101 B32: attention kernel times after the backward pipelining
set -o pipefail
mkdir -p gpurun_out/az
export TMPDIR=/tmp
(cd /tmp && timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/az/prof -- python $GRAFT_REPO_ROOT/bench.py --task instance --model danet_resnet101 --crop 512 --in-channels 4 --batch 32 --steps 10 --warmup 2 > $GRAFT_REPO_ROOT/gpurun_out/az/prof.log 2>&1) &&
python tools/summarize_prof.py $(find gpurun_out/az/prof -name "*kernel_stats.csv" | head -1) 12 "DANet-R101 512 B32 instance task, v35 (2 warmup + 10 steps; autotune launches of the first step included)" > gpurun_out/az/danet_b32_profile.md 2>&1
rm -rf gpurun_out/az/prof
head -14 gpurun_out/az/danet_b32_profile.md; grep -E "pam_|cam_" gpurun_out/az/danet_b32_profile.md
