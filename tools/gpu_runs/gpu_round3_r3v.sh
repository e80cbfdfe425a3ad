mkdir -p gpurun_out/r3v && cd $GRAFT_REPO_ROOT
run() { name=$1; shift; timeout -k 10 300 "$@" > gpurun_out/r3v/$name.log 2>&1; echo "$name rc=$?"; }
timeout -k 10 400 python -u -m pytest tests/test_gpu_runtime.py -x -q --timeout 200 --timeout-method thread -k "not multi_gpu" > gpurun_out/r3v/pytest.log 2>&1; tail -3 gpurun_out/r3v/pytest.log
DPAMD_AUX_STREAM=0 run r101_b256_noaux python bench.py --batch 256 --steps 10 --warmup 4
DPAMD_AUX_STREAM=0 run x65_b128_noaux python bench.py --model deeplabv3plus_xception --output-stride 8 --batch 128 --steps 6 --warmup 3
DPAMD_AUX_STREAM=0 run x65_b64_noaux python bench.py --model deeplabv3plus_xception --output-stride 8 --batch 64 --steps 8 --warmup 4
grep -o "\"value\": [0-9.]*\|hbm_[a-z_]*\": [0-9.]*\|alloc_retries\": [0-9]*" gpurun_out/r3v/*.log
