# graph-step mode: its tests, then eager vs replay at small batches
mkdir -p gpurun_out/r6b
timeout -k 10 400 python -u -m pytest tests/test_gpu_hardening.py -x -q --timeout 200 --timeout-method thread -k "graph_step" > gpurun_out/r6b/pytest.log 2>&1 || { grep -n "^E \|Error" gpurun_out/r6b/pytest.log | head -20 | cut -c1-300; tail -5 gpurun_out/r6b/pytest.log; exit 1; }
tail -1 gpurun_out/r6b/pytest.log
for b in 8 16 32 64; do
  timeout -k 10 300 python bench.py --batch $b --steps 30 --warmup 10 > gpurun_out/r6b/eager_b$b.log 2>&1 || exit 1
  timeout -k 10 300 python bench.py --batch $b --steps 30 --warmup 10 --graph-step > gpurun_out/r6b/graph_b$b.log 2>&1 || { tail -5 gpurun_out/r6b/graph_b$b.log; exit 1; }
done
grep -o '"value": [0-9.]*\|"final_loss": [0-9.]*' gpurun_out/r6b/*_b*.log
