# zero-step skipping in the weight-gradient kernel: numerics first, then same-box A/B against the previous library
mkdir -p gpurun_out/r5q
timeout -k 10 400 python -u -m pytest tests/test_gpu_kernels.py -x -q --timeout 120 --timeout-method thread -k "wgrad" > gpurun_out/r5q/pytest_wgrad.log 2>&1 || { tail -30 gpurun_out/r5q/pytest_wgrad.log | cut -c1-200; exit 1; }
tail -1 gpurun_out/r5q/pytest_wgrad.log
timeout -k 10 200 python tools/bench_conv.py --batch 256 --filter aspp > gpurun_out/r5q/aspp_new.log 2>&1 || exit 1
DPAMD_HIP_LIB=$GRAFT_REPO_ROOT/distributedpytorch_amd/csrc/build/libdpamd_prev.so timeout -k 10 200 python tools/bench_conv.py --batch 256 --filter aspp > gpurun_out/r5q/aspp_prev.log 2>&1 || exit 1
grep "aspp" gpurun_out/r5q/aspp_new.log | cut -c1-120; echo "-- previous library"; grep "aspp" gpurun_out/r5q/aspp_prev.log | cut -c1-120
