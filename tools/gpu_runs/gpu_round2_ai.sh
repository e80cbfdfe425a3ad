#!/bin/bash
# X65 os8 B64: what the weight-gradient stream costs the step (diagnostic knobs: invalid training, timing only)
set -o pipefail
mkdir -p gpurun_out/ai
X="python bench.py --model deeplabv3plus_xception65 --output-stride 8 --batch 64 --steps 10 --warmup 3"
timeout -k 10 300 $X > gpurun_out/ai/x65_base.log 2>&1 &&
DPAMD_SKIP_DW_WGRAD=1 timeout -k 10 300 $X > gpurun_out/ai/x65_skip_dw_wgrad.log 2>&1 &&
DPAMD_SKIP_DW_WGRAD=1 DPAMD_SKIP_WGRAD=1 timeout -k 10 300 $X > gpurun_out/ai/x65_skip_all_wgrad.log 2>&1
for f in base skip_dw_wgrad skip_all_wgrad; do echo $f; tail -1 gpurun_out/ai/x65_$f.log | cut -c1-220; done
