#!/bin/bash
# calibration of the X65 model-step test: residual-branch gain vs spread of the median gradient cosine
set -o pipefail
mkdir -p gpurun_out/bi
for g in 0.08 0.04 0.02; do for i in 1 2 3; do
  DPAMD_TEST_RES_GAIN=$g timeout -k 10 200 python -u -m pytest tests/test_gpu_hardening.py -m gpu -q -s -k "model_training_step and xception" --timeout 120 --timeout-method thread > gpurun_out/bi/g${g}_$i.log 2>&1; grep "^deeplab.*median cos" gpurun_out/bi/g${g}_$i.log | sed "s/^/gain $g  /"
done; done
exit 0
