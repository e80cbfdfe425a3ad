#!/bin/bash
# conv K loop: LDS-DMA pieces spread over the MFMA groups ONLY in rings of three stages (-DDP_SPREAD_DMA=2), every tile
# choice re-measured in the run (packaged table off) so that 3-stage tiles can win where they now do; same box, alternating
set -o pipefail
mkdir -p gpurun_out/cd
export TMPDIR=/tmp DPAMD_HEARTBEAT_S=45 DPAMD_CONV_TUNE_TABLE=0
R="python bench.py --batch 128 --steps 20 --warmup 5"
V=$GRAFT_REPO_ROOT/distributedpytorch_amd/csrc/build/libdpamd_spread2.so
DPAMD_HIP_LIB=$V timeout -k 10 400 $R > gpurun_out/cd/r101_spread2_a.log 2>&1 &&
timeout -k 10 400 $R > gpurun_out/cd/r101_default_a.log 2>&1 &&
DPAMD_HIP_LIB=$V timeout -k 10 400 $R > gpurun_out/cd/r101_spread2_b.log 2>&1 &&
timeout -k 10 400 $R > gpurun_out/cd/r101_default_b.log 2>&1
for f in r101_spread2_a r101_default_a r101_spread2_b r101_default_b; do echo $f; tail -1 gpurun_out/cd/$f.log | cut -c1-200; done
