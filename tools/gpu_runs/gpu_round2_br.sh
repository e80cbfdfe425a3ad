#!/bin/bash
# Xception-65 OS8 at per-GPU batch 96 / 128 (64 is the measured configuration)
set -o pipefail
mkdir -p gpurun_out/br
export TMPDIR=/tmp DPAMD_HEARTBEAT_S=45
for b in 96 128; do
DPAMD_CONV_TUNE_FILE=$GRAFT_REPO_ROOT/gpurun_out/br/x65_b${b}_tune.json timeout -k 10 500 python bench.py --model deeplabv3plus_xception65 --output-stride 8 --batch $b --steps 8 --warmup 3 > gpurun_out/br/x65_b$b.log 2>&1; tail -1 gpurun_out/br/x65_b$b.log | cut -c1-220
python - <<'PY'
import torch
print("peak GiB n/a (separate process)")
PY
done
