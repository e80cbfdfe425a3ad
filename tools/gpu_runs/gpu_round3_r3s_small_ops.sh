mkdir -p gpurun_out/r3s && cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && \
timeout -k 10 200 python tools/count_small_ops.py deeplabv3plus_xception --os8 > gpurun_out/r3s/small_ops_x65.log 2>&1 && \
timeout -k 10 200 python tools/count_small_ops.py danet_resnet101 > gpurun_out/r3s/small_ops_danet.log 2>&1 && \
timeout -k 10 200 python tools/count_small_ops.py deeplabv3plus_resnet101 > gpurun_out/r3s/small_ops_r101.log 2>&1 && \
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r3s/prof_x65_b128 -o p -- python3 bench.py --model deeplabv3plus_xception --output-stride 8 --batch 128 --steps 3 --warmup 2 > gpurun_out/r3s/prof_x65_b128.log 2>&1 ; rm -f gpurun_out/r3s/*/*kernel_trace.csv; tail -3 gpurun_out/r3s/*.log | cut -c1-300
