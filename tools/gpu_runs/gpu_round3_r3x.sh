mkdir -p gpurun_out/r3x && cd $GRAFT_REPO_ROOT
run() { name=$1; shift; timeout -k 10 300 "$@" > gpurun_out/r3x/$name.log 2>&1; echo "$name rc=$?"; }
timeout -k 10 700 python -u -m pytest tests -m gpu -x -q --timeout 200 --timeout-method thread > gpurun_out/r3x/pytest.log 2>&1; tail -3 gpurun_out/r3x/pytest.log
run r101_b256 python bench.py --batch 256 --steps 12 --warmup 5
run r101_b384 python bench.py --batch 384 --steps 10 --warmup 4
run r101_b512 python bench.py --batch 512 --steps 8 --warmup 4
run x65_b64 python bench.py --model deeplabv3plus_xception --output-stride 8 --batch 64 --steps 10 --warmup 4
run x65_b128 python bench.py --model deeplabv3plus_xception --output-stride 8 --batch 128 --steps 8 --warmup 4
run danet_b32 python bench.py --model danet_resnet101 --task instance --crop 512 --batch 32 --steps 15 --warmup 6
run danet_b64 python bench.py --model danet_resnet101 --task instance --crop 512 --batch 64 --steps 10 --warmup 5
grep -o "\"value\": [0-9.]*\|hbm_[a-z_]*\": [0-9.]*\|alloc_retries\": [0-9]*" gpurun_out/r3x/*.log
