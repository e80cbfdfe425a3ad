#!/bin/bash
# start-time stagger (de-phasing) sweep with phase stamps
set -o pipefail
mkdir -p gpurun_out/ac
timeout -k 10 250 python -u tools/stamp_conv.py --batch 128 --cfgs 16,9 --modes plain,stats,bn --stagger "0,0;4,0;8,0;11,0;16,0;8,2;11,3" 256:1024:33:1 1024:256:33:1 > gpurun_out/ac/stamp_r101.log 2>&1 &&
timeout -k 10 250 python -u tools/stamp_conv.py --batch 64 --cfgs 5 --modes plain,stats,resaff --stagger "0,0;0,6;0,12;0,20" 728:728:65:1 > gpurun_out/ac/stamp_x65.log 2>&1
cat gpurun_out/ac/stamp_r101.log gpurun_out/ac/stamp_x65.log
