#!/bin/bash
# DANet-R101 512 B16 instance task: kernel breakdown with the autotune launches amortised over 20 steps
set -o pipefail
mkdir -p gpurun_out/an
export TMPDIR=/tmp
(cd /tmp && timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/an/prof -- python $GRAFT_REPO_ROOT/bench.py --task instance --model danet_resnet101 --crop 512 --in-channels 4 --batch 16 --steps 20 --warmup 2 > $GRAFT_REPO_ROOT/gpurun_out/an/prof.log 2>&1) &&
python tools/summarize_prof.py $(find gpurun_out/an/prof -name "*kernel_stats.csv" | head -1) 22 "DANet-R101 512 B16 instance task, v32 (2 warmup + 20 steps; autotune launches of the first step included)" > gpurun_out/an/danet_profile.md 2>&1
rm -rf gpurun_out/an/prof
head -50 gpurun_out/an/danet_profile.md
