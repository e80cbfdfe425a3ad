mkdir -p gpurun_out/r4b && cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 200 --timeout-method thread > gpurun_out/r4b/pytest.log 2>&1; tail -3 gpurun_out/r4b/pytest.log
timeout -k 10 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/r4b/smoke.log 2>&1; tail -2 gpurun_out/r4b/smoke.log
timeout -k 10 300 python tools/count_small_ops.py deeplabv3plus_resnet101 > gpurun_out/r4b/small_ops_r101.log 2>&1
grep "aten::\(zero_\|fill_\|copy_\|add_\|mm\|zeros\|sum\|mul\) " gpurun_out/r4b/small_ops_r101.log | head
