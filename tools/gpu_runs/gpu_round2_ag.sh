#!/bin/bash
# persistent 256x256 (cfg 21) vs cfg 5: equality + time
set -o pipefail
mkdir -p gpurun_out/ag
timeout -k 10 300 python -u tools/check_cfg.py --batch 64 --a 5 --b 21 728:728:65:1 1536:1536:65:1 256:256:65:3:2 > gpurun_out/ag/persist.log 2>&1; rc=$?
cat gpurun_out/ag/persist.log; exit $rc
