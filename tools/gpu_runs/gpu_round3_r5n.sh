# PMC counters of the final tree's R101 step at B128, per kernel (three passes, each its own run): MFMA busy, VALU, LDS,
# waits; L2 hits / misses / fabric reads
mkdir -p gpurun_out/r5n && cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
A="bench.py --batch 128 --steps 2 --warmup 1"
timeout -s KILL 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT --kernel-trace --output-format csv -d gpurun_out/r5n/pmc_sq -- python3 $A > gpurun_out/r5n/run_sq.log 2>&1 || { tail -5 gpurun_out/r5n/run_sq.log; exit 1; }
timeout -s KILL 300 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum --kernel-trace --output-format csv -d gpurun_out/r5n/pmc_tcc -- python3 $A > gpurun_out/r5n/run_tcc.log 2>&1 || { tail -5 gpurun_out/r5n/run_tcc.log; exit 1; }
python tools/pmc_summary.py gpurun_out/r5n/pmc_sq gpurun_out/r5n/pmc_tcc > gpurun_out/r5n/pmc_r101_b128.txt 2>&1; rm -rf gpurun_out/r5n/pmc_sq gpurun_out/r5n/pmc_tcc; wc -l gpurun_out/r5n/pmc_r101_b128.txt; head -30 gpurun_out/r5n/pmc_r101_b128.txt | cut -c1-150
