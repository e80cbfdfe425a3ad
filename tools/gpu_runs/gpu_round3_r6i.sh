mkdir -p gpurun_out/r6i
timeout -k 10 400 python -u -m pytest tests/test_gpu_p2p.py tests/test_gpu_hardening.py -x -q --timeout 200 --timeout-method thread -k "graph_step" > gpurun_out/r6i/pytest.log 2>&1; rc=$?; grep -n "^E  \|Error" gpurun_out/r6i/pytest.log | head -12 | cut -c1-300; tail -2 gpurun_out/r6i/pytest.log | cut -c1-200
[ $rc -eq 0 ] || exit 1
DPAMD_DIST_BACKEND=gloo timeout -k 10 300 python bench.py --gpus 2 --batch 8 --steps 16 --warmup 8 > gpurun_out/r6i/two_ranks_eager.log 2>&1 || { tail -5 gpurun_out/r6i/two_ranks_eager.log; exit 1; }
DPAMD_DIST_BACKEND=gloo timeout -k 10 300 python bench.py --gpus 2 --batch 8 --steps 16 --warmup 8 --graph-step > gpurun_out/r6i/two_ranks_graph.log 2>&1 || { tail -8 gpurun_out/r6i/two_ranks_graph.log; exit 1; }
DPAMD_DDP_FORCE=1 timeout -k 10 300 python bench.py --batch 16 --steps 20 --warmup 8 --graph-step > gpurun_out/r6i/rccl_1rank_graph_b16.log 2>&1 || { tail -8 gpurun_out/r6i/rccl_1rank_graph_b16.log; exit 1; }
DPAMD_DDP_FORCE=1 timeout -k 10 300 python bench.py --batch 16 --steps 20 --warmup 8 > gpurun_out/r6i/rccl_1rank_eager_b16.log 2>&1 || exit 1
grep -o '"value": [0-9.]*\|"graphs_captured": [0-9]*' gpurun_out/r6i/*.log
