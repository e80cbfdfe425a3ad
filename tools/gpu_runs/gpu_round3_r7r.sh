# after the dead-code removal (library rebuilt): full GPU suite,
# smoke, default benches of the three families
mkdir -p gpurun_out/r7r
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread > gpurun_out/r7r/pytest.log 2>&1 || { tail -30 gpurun_out/r7r/pytest.log; exit 1; }
tail -2 gpurun_out/r7r/pytest.log
timeout -k 10 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/r7r/smoke.log 2>&1 || { tail -20 gpurun_out/r7r/smoke.log; exit 1; }
timeout -k 10 300 python bench.py > gpurun_out/r7r/r101_b256.log 2>&1 && \
timeout -k 10 300 python bench.py --model deeplabv3plus_xception --output-stride 8 --steps 12 --warmup 5 > gpurun_out/r7r/x65_b128.log 2>&1 && \
timeout -k 10 300 python bench.py --model danet_resnet101 --task instance --crop 512 --steps 20 --warmup 6 > gpurun_out/r7r/danet_b64.log 2>&1 && \
DPAMD_DDP_FORCE=1 timeout -k 10 300 python bench.py --sync-bn > gpurun_out/r7r/r101_b256_syncbn_forced.log 2>&1; rc=$?
tail -1 gpurun_out/r7r/smoke.log; grep -o '"value": [0-9.]*\|"tiles_autotuned_in_run": [0-9]*' gpurun_out/r7r/*.log; exit $rc
