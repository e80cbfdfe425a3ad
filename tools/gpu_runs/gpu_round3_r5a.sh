mkdir -p gpurun_out/r5a
timeout -k 10 600 python -u -m pytest tests/test_gpu_p2p.py -x -q --timeout 300 --timeout-method thread > gpurun_out/r5a/pytest.log 2>&1; rc=$?; tail -4 gpurun_out/r5a/pytest.log | cut -c1-250; exit $rc
