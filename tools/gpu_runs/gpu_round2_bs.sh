#!/bin/bash
# maxpool / bilinear backward with their loads batched: tests and kernel times inside an R101 B128 step
set -o pipefail
mkdir -p gpurun_out/bs
export TMPDIR=/tmp DPAMD_HEARTBEAT_S=45
timeout -k 10 400 python -u -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "pool or bilinear or upsample or resize" --timeout 120 --timeout-method thread > gpurun_out/bs/tests.log 2>&1 || { tail -40 gpurun_out/bs/tests.log; exit 1; }
tail -1 gpurun_out/bs/tests.log
(cd /tmp && timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/bs/prof -- python $GRAFT_REPO_ROOT/bench.py --batch 128 --steps 4 --warmup 1 > $GRAFT_REPO_ROOT/gpurun_out/bs/prof.log 2>&1) &&
python tools/summarize_prof.py $(find gpurun_out/bs/prof -name "*kernel_stats.csv" | head -1) 5 "R101 B128 (maxpool/bilinear backward batched loads)" > gpurun_out/bs/profile.md 2>&1
rm -rf gpurun_out/bs/prof
grep -i "maxpool\|bilinear\|upsample_ce" gpurun_out/bs/profile.md | cut -c1-120
