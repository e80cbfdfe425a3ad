#!/bin/bash
# queue idle time of an Xception-65 and an R101 step from the kernel trace
set -o pipefail
mkdir -p gpurun_out/be
export TMPDIR=/tmp DPAMD_HEARTBEAT_S=45
(cd /tmp && timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/be/prof_x65 -- python $GRAFT_REPO_ROOT/bench.py --model deeplabv3plus_xception65 --output-stride 8 --batch 64 --steps 4 --warmup 2 > $GRAFT_REPO_ROOT/gpurun_out/be/prof_x65.log 2>&1) &&
python tools/trace_gaps.py $(find gpurun_out/be/prof_x65 -name "*kernel_trace.csv" | head -1) 0.5 > gpurun_out/be/x65_gaps.txt 2>&1 &&
(cd /tmp && timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/be/prof_r101 -- python $GRAFT_REPO_ROOT/bench.py --batch 128 --steps 4 --warmup 2 > $GRAFT_REPO_ROOT/gpurun_out/be/prof_r101.log 2>&1) &&
python tools/trace_gaps.py $(find gpurun_out/be/prof_r101 -name "*kernel_trace.csv" | head -1) 0.5 > gpurun_out/be/r101_gaps.txt 2>&1
head -3 $(find gpurun_out/be/prof_x65 -name "*kernel_trace.csv" | head -1) | cut -c1-400
rm -rf gpurun_out/be/prof_x65 gpurun_out/be/prof_r101
cat gpurun_out/be/x65_gaps.txt; cat gpurun_out/be/r101_gaps.txt
