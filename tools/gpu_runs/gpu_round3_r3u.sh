mkdir -p gpurun_out/r3u && cd $GRAFT_REPO_ROOT && export DPAMD_CONV_TUNE_FILE=$GRAFT_REPO_ROOT/gpurun_out/r3u/tune.json
run() { name=$1; shift; timeout -k 10 300 "$@" > gpurun_out/r3u/$name.log 2>&1; echo "$name rc=$?"; }
X65="python bench.py --model deeplabv3plus_xception --output-stride 8 --batch 128 --steps 6 --warmup 3"
PYTORCH_HIP_ALLOC_CONF=max_split_size_mb:64 run x65_b128_split64 $X65
PYTORCH_HIP_ALLOC_CONF=max_split_size_mb:512 run x65_b128_split512 $X65
PYTORCH_HIP_ALLOC_CONF=garbage_collection_threshold:0.7 run x65_b128_gc07 $X65
run x65_b96 python bench.py --model deeplabv3plus_xception --output-stride 8 --batch 96 --steps 8 --warmup 4
run r101_b320 python bench.py --batch 320 --steps 12 --warmup 5
run r101_b384 python bench.py --batch 384 --steps 12 --warmup 5
run r101_b256 python bench.py --batch 256 --steps 12 --warmup 5
PYTORCH_HIP_ALLOC_CONF=max_split_size_mb:64 run r101_b448_split64 python bench.py --batch 448 --steps 6 --warmup 3
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r3u/prof_b256 -o p -- python3 bench.py --steps 4 --warmup 3 --batch 256 > gpurun_out/r3u/prof_b256.log 2>&1
rm -f gpurun_out/r3u/*/*kernel_trace.csv
grep -o "\"value\": [0-9.]*\|hbm_[a-z_]*\": [0-9.]*\|alloc_retries\": [0-9]*" gpurun_out/r3u/*.log
