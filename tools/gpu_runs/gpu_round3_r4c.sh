# library-GEMM yardstick for the 1x1 layers (R101 B256, X65 OS8 B128) + the R101 B128 kernel profile of the final tree
mkdir -p gpurun_out/r4c && cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && \
timeout -k 10 280 python -u tools/bench_conv.py --batch 256 --lib-gemm > gpurun_out/r4c/libgemm_r101_b256.log 2>&1 && \
timeout -k 10 200 python -u tools/bench_conv.py --batch 128 --net x65 --lib-gemm > gpurun_out/r4c/libgemm_x65_b128.log 2>&1 && \
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r4c/prof_b128 -o p -- python3 bench.py --batch 128 --steps 4 --warmup 3 > gpurun_out/r4c/prof_b128.log 2>&1 ; rc=$?; rm -f gpurun_out/r4c/*/*kernel_trace.csv; tail -3 gpurun_out/r4c/libgemm_r101_b256.log; tail -3 gpurun_out/r4c/libgemm_x65_b128.log; exit $rc
