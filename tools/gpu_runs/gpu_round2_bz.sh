#!/bin/bash
# conv fwd/dgrad K loop: LDS-DMA pieces of the next stage spread over the MFMA groups
# tests, block stamps, same-box A/B of R101 B128 and X65 against the previous library
set -o pipefail
mkdir -p gpurun_out/bz
export TMPDIR=/tmp DPAMD_HEARTBEAT_S=45
timeout -k 10 500 python -u -m pytest tests -m gpu -x -q  --timeout 120 --timeout-method thread > gpurun_out/bz/tests.log 2>&1 || { tail -40 gpurun_out/bz/tests.log; exit 1; }
tail -2 gpurun_out/bz/tests.log
timeout -k 10 300 true
R="python bench.py --batch 128 --steps 20 --warmup 5"
X="python bench.py --model deeplabv3plus_xception65 --output-stride 8 --batch 64 --steps 10 --warmup 3"
PREV=$GRAFT_REPO_ROOT/distributedpytorch_amd/csrc/build/libdpamd_prev.so
timeout -k 10 300 $R > gpurun_out/bz/r101_new_a.log 2>&1 &&
DPAMD_HIP_LIB=$PREV timeout -k 10 300 $R > gpurun_out/bz/r101_prev_a.log 2>&1 &&
timeout -k 10 300 $R > gpurun_out/bz/r101_new_b.log 2>&1 &&
DPAMD_HIP_LIB=$PREV timeout -k 10 300 $R > gpurun_out/bz/r101_prev_b.log 2>&1 &&
timeout -k 10 300 $X > gpurun_out/bz/x65_new_a.log 2>&1 &&
DPAMD_HIP_LIB=$PREV timeout -k 10 300 $X > gpurun_out/bz/x65_prev_a.log 2>&1
for f in r101_new_a r101_prev_a r101_new_b r101_prev_b x65_new_a x65_prev_a; do echo $f; tail -1 gpurun_out/bz/$f.log | cut -c1-200; done
