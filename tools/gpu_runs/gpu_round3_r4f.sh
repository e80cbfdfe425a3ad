# what the statistics epilogue costs on the layers where the library GEMM is ahead
mkdir -p gpurun_out/r4f && cd $GRAFT_REPO_ROOT && \
timeout -k 10 200 python -u tools/bench_conv.py --batch 256 --lib-gemm --filter l3 > gpurun_out/r4f/l3.log 2>&1 && \
timeout -k 10 200 python -u tools/bench_conv.py --batch 256 --lib-gemm --filter l4 > gpurun_out/r4f/l4.log 2>&1 && \
timeout -k 10 200 python -u tools/bench_conv.py --batch 128 --net x65 --lib-gemm --filter pw768 > gpurun_out/r4f/x65.log 2>&1; rc=$?; cat gpurun_out/r4f/*.log | grep -v amdgpu; exit $rc
