mkdir -p gpurun_out/r3q && export DPAMD_CONV_TUNE_FILE=$GRAFT_REPO_ROOT/gpurun_out/r3q/tune.json && cp gpurun_out/r3p_tune.json $DPAMD_CONV_TUNE_FILE 2>/dev/null; \
timeout -k 10 300 python bench.py --steps 12 --warmup 5 --batch 256 > gpurun_out/r3q/b256.log 2>&1 && \
timeout -k 10 300 python bench.py --steps 12 --warmup 5 --batch 384 > gpurun_out/r3q/b384.log 2>&1 && \
timeout -k 10 300 python bench.py --steps 12 --warmup 5 --batch 256 > gpurun_out/r3q/b256_2.log 2>&1 && \
timeout -k 10 300 python bench.py --steps 12 --warmup 5 --batch 384 > gpurun_out/r3q/b384_2.log 2>&1 && \
timeout -k 10 300 python bench.py --steps 12 --warmup 5 --batch 320 > gpurun_out/r3q/b320.log 2>&1 && \
timeout -k 10 300 python bench.py --steps 12 --warmup 5 --batch 448 > gpurun_out/r3q/b448.log 2>&1 && \
MIOPEN_FIND_MODE=FAST timeout -k 10 500 python bench.py --impl stock --steps 6 --warmup 2 --batch 256 > gpurun_out/r3q/stock_b256_fast.log 2>&1 ; grep -o "\"value\": [0-9.]*\|hbm_peak_gib\": [0-9.]*" gpurun_out/r3q/*.log
