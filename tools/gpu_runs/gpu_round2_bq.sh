#!/bin/bash
# bench.py default (per-GPU batch 256, packaged tile table) and batch 128 on the same box; chunk tests
set -o pipefail
mkdir -p gpurun_out/bq
export TMPDIR=/tmp DPAMD_HEARTBEAT_S=45
timeout -k 10 500 python -u -m pytest tests/test_gpu_hardening.py -m gpu -x -q --timeout 180 --timeout-method thread > gpurun_out/bq/tests.log 2>&1 || { tail -40 gpurun_out/bq/tests.log; exit 1; }
tail -1 gpurun_out/bq/tests.log
timeout -k 10 500 python bench.py > gpurun_out/bq/r101_default.log 2>&1 &&
timeout -k 10 500 python bench.py --batch 128 > gpurun_out/bq/r101_b128.log 2>&1 &&
timeout -k 10 500 python bench.py > gpurun_out/bq/r101_default_b.log 2>&1 &&
(cd /tmp && timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/bq/prof_r101 -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 > $GRAFT_REPO_ROOT/gpurun_out/bq/prof_r101.log 2>&1) &&
python tools/summarize_prof.py $(find gpurun_out/bq/prof_r101 -name "*kernel_stats.csv" | head -1) 4 "DeepLabv3+ R101 513x513 B256, v38 (1 warmup + 3 steps)" > gpurun_out/bq/r101_b256_profile.md 2>&1
rm -rf gpurun_out/bq/prof_r101
for f in r101_default r101_b128 r101_default_b; do echo $f; tail -1 gpurun_out/bq/$f.log | cut -c1-1500; done
head -14 gpurun_out/bq/r101_b256_profile.md
