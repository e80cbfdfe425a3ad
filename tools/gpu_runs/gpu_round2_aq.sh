#!/bin/bash
# X65 step: depthwise generation 4 vs 6 (matrix-core forward / plain data gradient only) vs 7 (+ fused data gradient)
set -o pipefail
mkdir -p gpurun_out/aq
X="python bench.py --model deeplabv3plus_xception65 --output-stride 8 --batch 64 --steps 10 --warmup 3"
for g in 4 6 7 4 6; do
  DPAMD_DWCONV_GEN=$g timeout -k 10 300 $X > gpurun_out/aq/x65_gen$g.log 2>&1 || exit 1
  echo gen$g; tail -1 gpurun_out/aq/x65_gen$g.log | cut -c1-200
done
