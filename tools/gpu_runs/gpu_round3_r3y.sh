mkdir -p gpurun_out/r3y && cd $GRAFT_REPO_ROOT
run() { name=$1; shift; timeout -k 10 300 "$@" > gpurun_out/r3y/$name.log 2>&1; echo "$name rc=$?"; }
timeout -k 10 300 python -u -m pytest tests/test_gpu_p2p.py -x -q --timeout 200 --timeout-method thread > gpurun_out/r3y/pytest.log 2>&1; tail -3 gpurun_out/r3y/pytest.log
for lag in 4 16 64 2 8 4; do
DPAMD_AUX_LAG=$lag run r101_b256_lag${lag}_$RANDOM python bench.py --batch 256 --steps 12 --warmup 5
done
DPAMD_AUX_STREAM=0 run r101_b256_noaux python bench.py --batch 256 --steps 12 --warmup 5
grep -o "\"value\": [0-9.]*\|hbm_reserved[a-z_]*\": [0-9.]*" gpurun_out/r3y/*.log
