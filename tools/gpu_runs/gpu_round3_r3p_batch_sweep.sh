mkdir -p gpurun_out/r3p && export DPAMD_CONV_TUNE_FILE=$GRAFT_REPO_ROOT/gpurun_out/r3p/tune.json && \
timeout -k 10 300 python bench.py --steps 8 --warmup 4 --batch 384 > gpurun_out/r3p/b384.log 2>&1 && \
timeout -k 10 300 python bench.py --steps 8 --warmup 4 --batch 384 > gpurun_out/r3p/b384_2.log 2>&1 && \
timeout -k 10 300 python bench.py --steps 6 --warmup 3 --batch 512 > gpurun_out/r3p/b512.log 2>&1 && \
timeout -k 10 300 python bench.py --steps 6 --warmup 3 --batch 512 > gpurun_out/r3p/b512_2.log 2>&1 ; grep -o "\"value\": [0-9.]*" gpurun_out/r3p/*.log; tail -3 gpurun_out/r3p/b512_2.log | cut -c1-600
