#!/bin/bash
# v38 validation: full GPU tests, benches (R101 B128, X65 os8 B64, DANet instance B16), rocprof breakdown of R101 B128 and X65
set -o pipefail
mkdir -p gpurun_out/ca
export TMPDIR=/tmp DPAMD_HEARTBEAT_S=45
timeout -k 10 600 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread > gpurun_out/ca/tests.log 2>&1 || { tail -30 gpurun_out/ca/tests.log; exit 1; }
tail -2 gpurun_out/ca/tests.log
timeout -k 10 300 python bench.py --steps 20 --warmup 5 > gpurun_out/ca/r101_b128.log 2>&1 &&
timeout -k 10 300 python bench.py --model deeplabv3plus_xception65 --output-stride 8 --batch 64 --steps 10 --warmup 3 > gpurun_out/ca/x65_b64.log 2>&1 &&
timeout -k 10 300 python bench.py --task instance --model danet_resnet101 --crop 512 --in-channels 4 --batch 16 --steps 10 --warmup 3 > gpurun_out/ca/danet_b16.log 2>&1 &&
timeout -k 10 300 python bench.py --task instance --model danet_resnet101 --crop 512 --in-channels 4 --batch 32 --steps 10 --warmup 3 > gpurun_out/ca/danet_b32.log 2>&1 &&
(cd /tmp && timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/ca/prof_r101 -- python $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 1 > $GRAFT_REPO_ROOT/gpurun_out/ca/prof_r101.log 2>&1) &&
python tools/summarize_prof.py $(find gpurun_out/ca/prof_r101 -name "*kernel_stats.csv" | head -1) 5 "DeepLabv3+ R101 513x513 B256, v38 (1 warmup + 4 steps)" > gpurun_out/ca/r101_b128_profile.md 2>&1 &&
(cd /tmp && timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/ca/prof_x65 -- python $GRAFT_REPO_ROOT/bench.py --model deeplabv3plus_xception65 --output-stride 8 --batch 64 --steps 4 --warmup 1 > $GRAFT_REPO_ROOT/gpurun_out/ca/prof_x65.log 2>&1) &&
python tools/summarize_prof.py $(find gpurun_out/ca/prof_x65 -name "*kernel_stats.csv" | head -1) 5 "Xception-65 OS8 B64, v38 (1 warmup + 4 steps)" > gpurun_out/ca/x65_b64_profile.md 2>&1
rm -rf gpurun_out/ca/prof_r101 gpurun_out/ca/prof_x65
for f in r101_b128 x65_b64 danet_b16 danet_b32; do tail -1 gpurun_out/ca/$f.log | cut -c1-210; done
head -14 gpurun_out/ca/r101_b128_profile.md; head -14 gpurun_out/ca/x65_b64_profile.md
