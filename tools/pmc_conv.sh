#!/bin/bash
# usage: tools/pmc_conv.sh <tag> <one_conv args...>   -> gpurun_out/pmc_<tag>_{sq,sq2,tcc}/
set -e
tag=$1; shift
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT --kernel-trace --output-format csv -d $root/gpurun_out/pmc_${tag}_sq -- python $root/tools/one_conv.py "$@" > $root/gpurun_out/pmc_${tag}_sq.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_MFMA SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $root/gpurun_out/pmc_${tag}_sq2 -- python $root/tools/one_conv.py "$@" > $root/gpurun_out/pmc_${tag}_sq2.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_REQ_sum --kernel-trace --output-format csv -d $root/gpurun_out/pmc_${tag}_tcc -- python $root/tools/one_conv.py "$@" > $root/gpurun_out/pmc_${tag}_tcc.log 2>&1
echo pmc $tag done
