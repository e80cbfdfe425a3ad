mkdir -p gpurun_out/r3w && cd $GRAFT_REPO_ROOT
run() { name=$1; shift; timeout -k 10 300 "$@" > gpurun_out/r3w/$name.log 2>&1; echo "$name rc=$?"; }
for lag in 2 4 8 16; do
DPAMD_AUX_LAG=$lag run r101_b256_lag$lag python bench.py --batch 256 --steps 10 --warmup 4
done
DPAMD_AUX_LAG=0 run r101_b256_lag0 python bench.py --batch 256 --steps 10 --warmup 4
for lag in 2 4 8; do
DPAMD_AUX_LAG=$lag run x65_b128_lag$lag python bench.py --model deeplabv3plus_xception --output-stride 8 --batch 128 --steps 6 --warmup 3
done
DPAMD_AUX_LAG=4 run x65_b64_lag4 python bench.py --model deeplabv3plus_xception --output-stride 8 --batch 64 --steps 8 --warmup 4
DPAMD_AUX_LAG=4 run r101_b384_lag4 python bench.py --batch 384 --steps 10 --warmup 4
DPAMD_AUX_LAG=4 run r101_b512_lag4 python bench.py --batch 512 --steps 6 --warmup 3
grep -o "\"value\": [0-9.]*\|hbm_reserved[a-z_]*\": [0-9.]*\|alloc_retries\": [0-9]*" gpurun_out/r3w/*.log
