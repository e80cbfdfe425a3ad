# same-box A/B of the rebuilt kernel library (dead kernel removed) against the previous build (DPAMD_HIP_LIB), interleaved
mkdir -p gpurun_out/r5p
for rep in 1 2; do
  timeout -k 10 300 python bench.py --steps 16 --warmup 6 > gpurun_out/r5p/new_$rep.log 2>&1 || exit 1
  DPAMD_HIP_LIB=$GRAFT_REPO_ROOT/distributedpytorch_amd/csrc/build/libdpamd_prev.so timeout -k 10 300 python bench.py --steps 16 --warmup 6 > gpurun_out/r5p/prev_$rep.log 2>&1 || exit 1
done
grep -o '"value": [0-9.]*\|"lib_hash": "[a-z0-9_.]*"' gpurun_out/r5p/*.log
