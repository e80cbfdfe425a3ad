mkdir -p gpurun_out/r4a && cd $GRAFT_REPO_ROOT && export DPAMD_CONV_TUNE_FILE=$GRAFT_REPO_ROOT/gpurun_out/r4a/tune.json
run() { name=$1; shift; timeout -k 10 300 "$@" > gpurun_out/r4a/$name.log 2>&1; echo "$name rc=$?"; }
run t_danet_b64 python bench.py --model danet_resnet101 --task instance --crop 512 --steps 6 --warmup 4
run t_x65_b128 python bench.py --model deeplabv3plus_xception --output-stride 8 --steps 6 --warmup 4
run t_x65_b96 python bench.py --model deeplabv3plus_xception --output-stride 8 --batch 96 --steps 6 --warmup 4
run t_r101_b256 python bench.py --steps 6 --warmup 4
run t_r101_b512 python bench.py --batch 512 --steps 6 --warmup 4
DPAMD_DDP_FORCE=1 run t_r101_b256_syncbn python bench.py --steps 6 --warmup 4 --sync-bn
DPAMD_DDP_FORCE=1 run t_x65_b128_syncbn python bench.py --model deeplabv3plus_xception --output-stride 8 --steps 6 --warmup 4 --sync-bn
run danet_b64 python bench.py --model danet_resnet101 --task instance --crop 512 --steps 12 --warmup 5
run x65_b128 python bench.py --model deeplabv3plus_xception --output-stride 8 --steps 10 --warmup 4
run r101_b256 python bench.py
grep -o "\"value\": [0-9.]*\|tiles_autotuned_in_run\": [0-9]*" gpurun_out/r4a/*.log
