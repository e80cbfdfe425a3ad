#!/bin/bash
# PMC counters of the weight-gradient kernel on the layer-3 3x3 256->256 problem, batch 128 (three passes, one counter group each)
set -o pipefail
mkdir -p gpurun_out/bv
bash tools/pmc_conv.sh wg33 --cin 256 --cout 256 --hw 33 --k 3 --batch 128 --pass wgrad --iters 4 > gpurun_out/bv/run.log 2>&1 || { tail -5 gpurun_out/bv/run.log; exit 1; }
python tools/pmc_summary.py gpurun_out/pmc_wg33_sq gpurun_out/pmc_wg33_sq2 gpurun_out/pmc_wg33_tcc > gpurun_out/bv/pmc_wgrad_l3_3x3_b128.txt 2>&1
rm -rf gpurun_out/pmc_wg33_sq gpurun_out/pmc_wg33_sq2 gpurun_out/pmc_wg33_tcc
cat gpurun_out/bv/pmc_wgrad_l3_3x3_b128.txt
