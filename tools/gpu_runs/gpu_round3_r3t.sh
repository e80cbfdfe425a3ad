mkdir -p gpurun_out/r3t && cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && export DPAMD_CONV_TUNE_FILE=$GRAFT_REPO_ROOT/gpurun_out/r3t/tune.json && \
timeout -k 10 600 python -u -m pytest tests/test_gpu_kernels.py tests/test_gpu_sepconv.py tests/test_gpu_attention.py tests/test_gpu_hardening.py -x -q --timeout 150 --timeout-method thread > gpurun_out/r3t/pytest.log 2>&1 && \
timeout -k 10 200 python bench.py --model deeplabv3plus_xception --output-stride 8 --batch 64 --steps 12 --warmup 5 > gpurun_out/r3t/x65_b64.log 2>&1 && \
timeout -k 10 200 python bench.py --model danet_resnet101 --task instance --crop 512 --batch 32 --steps 15 --warmup 6 > gpurun_out/r3t/danet_b32.log 2>&1 && \
timeout -k 10 200 python tools/count_small_ops.py deeplabv3plus_xception --os8 > gpurun_out/r3t/small_ops_x65.log 2>&1 && \
timeout -k 10 200 python tools/count_small_ops.py danet_resnet101 > gpurun_out/r3t/small_ops_danet.log 2>&1 && \
timeout -k 10 300 python bench.py --model deeplabv3plus_xception --output-stride 8 --batch 128 --steps 6 --warmup 3 > gpurun_out/r3t/x65_b128_default_alloc.log 2>&1 && \
PYTORCH_HIP_ALLOC_CONF=expandable_segments:True timeout -k 10 300 python bench.py --model deeplabv3plus_xception --output-stride 8 --batch 128 --steps 6 --warmup 3 > gpurun_out/r3t/x65_b128_expandable.log 2>&1 ; \
PYTORCH_HIP_ALLOC_CONF=expandable_segments:True timeout -k 10 300 python bench.py --batch 448 --steps 6 --warmup 3 > gpurun_out/r3t/r101_b448_expandable.log 2>&1 ; \
grep -o "\"value\": [0-9.]*\|hbm_[a-z_]*\": [0-9.]*\|alloc_retries\": [0-9]*" gpurun_out/r3t/*.log; tail -2 gpurun_out/r3t/pytest.log
