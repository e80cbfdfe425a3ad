# does replaying the step from a graph change anything at the bench default (B256)?  interleaved, same box
mkdir -p gpurun_out/r6q
for rep in 1 2; do
  timeout -k 10 300 python bench.py --steps 16 --warmup 6 > gpurun_out/r6q/eager_$rep.log 2>&1 || exit 1
  timeout -k 10 300 python bench.py --steps 16 --warmup 6 --graph-step > gpurun_out/r6q/graph_$rep.log 2>&1 || { tail -5 gpurun_out/r6q/graph_$rep.log; exit 1; }
done
timeout -k 10 300 python bench.py --batch 128 --steps 16 --warmup 6 > gpurun_out/r6q/b128_eager.log 2>&1 || exit 1
timeout -k 10 300 python bench.py --batch 128 --steps 16 --warmup 6 --graph-step > gpurun_out/r6q/b128_graph.log 2>&1 || exit 1
grep -o '"value": [0-9.]*\|"hbm_reserved_peak_gib": [0-9.]*' gpurun_out/r6q/*.log
