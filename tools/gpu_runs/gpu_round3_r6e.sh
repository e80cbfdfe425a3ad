# same-box A/B of the library with the device dropout word / device learning rate against the previous build
mkdir -p gpurun_out/r6e
for rep in 1 2 3; do
  timeout -k 10 300 python bench.py --steps 16 --warmup 6 > gpurun_out/r6e/new_$rep.log 2>&1 || exit 1
  DPAMD_HIP_LIB=$GRAFT_REPO_ROOT/distributedpytorch_amd/csrc/build/libdpamd_prev.so timeout -k 10 300 python bench.py --steps 16 --warmup 6 > gpurun_out/r6e/prev_$rep.log 2>&1 || exit 1
done
timeout -k 10 300 python bench.py --model deeplabv3plus_xception65 --output-stride 8 --steps 10 --warmup 4 > gpurun_out/r6e/x65_new.log 2>&1 || exit 1
DPAMD_HIP_LIB=$GRAFT_REPO_ROOT/distributedpytorch_amd/csrc/build/libdpamd_prev.so timeout -k 10 300 python bench.py --model deeplabv3plus_xception65 --output-stride 8 --steps 10 --warmup 4 > gpurun_out/r6e/x65_prev.log 2>&1 || exit 1
grep -o '"value": [0-9.]*' gpurun_out/r6e/*.log
