#!/bin/bash
# weight-gradient kernel (layer-3 3x3 256->256, batch 128): how much of a wave's time goes into issuing its LDS-DMA loads?
set -o pipefail
mkdir -p gpurun_out/bx
root=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM_RD SQ_INST_LEVEL_VMEM SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL --kernel-trace --output-format csv -d $root/gpurun_out/bx/pmc -- python $root/tools/one_conv.py --cin 256 --cout 256 --hw 33 --k 3 --batch 128 --pass wgrad --iters 4 > $root/gpurun_out/bx/run.log 2>&1 || { tail -5 $root/gpurun_out/bx/run.log; exit 1; }
cd $root && python tools/pmc_summary.py gpurun_out/bx/pmc > gpurun_out/bx/pmc_wgrad_vmem.txt 2>&1; rm -rf gpurun_out/bx/pmc; cat gpurun_out/bx/pmc_wgrad_vmem.txt
