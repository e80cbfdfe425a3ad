mkdir -p gpurun_out/r4h && cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && \
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r4h/prof -o p -- python3 tools/lib_gemm_probe.py > gpurun_out/r4h/run.log 2>&1; rc=$?; rm -f gpurun_out/r4h/prof/*kernel_trace.csv; cut -c1-400 gpurun_out/r4h/prof/p_kernel_stats.csv | head -20; exit $rc
