#!/bin/bash
# Xception-65 with the 728-channel flow stored 768 wide: GPU test, then three autotune runs for the packaged tile table
set -o pipefail
mkdir -p gpurun_out/bb
export TMPDIR=/tmp DPAMD_HEARTBEAT_S=45
timeout -k 10 300 python -u -m pytest tests/test_gpu_hardening.py tests/test_gpu_sepconv.py -m gpu -x -q --timeout 120 --timeout-method thread -k "xception or sepconv" > gpurun_out/bb/tests.log 2>&1 || { tail -30 gpurun_out/bb/tests.log; exit 1; }
tail -2 gpurun_out/bb/tests.log
for i in 1 2 3; do
  DPAMD_CONV_TUNE_FILE=$GRAFT_REPO_ROOT/gpurun_out/bb/x65_768_tune_$i.json timeout -k 10 400 python bench.py --model deeplabv3plus_xception65 --output-stride 8 --batch 64 --steps 8 --warmup 3 > gpurun_out/bb/x65_tune_$i.log 2>&1 || { tail -20 gpurun_out/bb/x65_tune_$i.log; exit 1; }
  tail -1 gpurun_out/bb/x65_tune_$i.log | cut -c1-200
done
