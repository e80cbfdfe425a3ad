#!/bin/bash
# spread of the X65 model-step test (median gradient cosine, fused vs plain) with the new and the previous library
set -o pipefail
mkdir -p gpurun_out/bh
PREV=$GRAFT_REPO_ROOT/distributedpytorch_amd/csrc/build/libdpamd_prev.so
T="python -u -m pytest tests/test_gpu_hardening.py -m gpu -q -s -k model_training_step_matches_fp32_cpu and xception --timeout 120 --timeout-method thread"
for i in 1 2 3; do
  timeout -k 10 200 python -u -m pytest tests/test_gpu_hardening.py -m gpu -q -s -k "model_training_step and xception" --timeout 120 --timeout-method thread > gpurun_out/bh/new_$i.log 2>&1; grep "median cos" gpurun_out/bh/new_$i.log | sed 's/^/new  /'
  DPAMD_HIP_LIB=$PREV timeout -k 10 200 python -u -m pytest tests/test_gpu_hardening.py -m gpu -q -s -k "model_training_step and xception" --timeout 120 --timeout-method thread > gpurun_out/bh/prev_$i.log 2>&1; grep "median cos" gpurun_out/bh/prev_$i.log | sed 's/^/prev /'
done
exit 0
