#!/bin/bash
set -u
O=gpurun_out/r7d; mkdir -p $O
timeout -k 10 500 python -u -m pytest tests/test_gpu_kernels.py tests/test_gpu_inference.py -x -q -k "bilinear or inference or predict or eval" --timeout 180 --timeout-method thread > $O/pytest.log 2>&1
rc=$?; tail -4 $O/pytest.log | cut -c1-300; [ $rc -ne 0 ] && exit $rc
for m in deeplabv3plus_resnet101 danet_resnet101; do
timeout -k 10 200 python -u tools/bench_eval.py $m 32 > $O/eval_${m}_b32.log 2>&1 || { tail -5 $O/eval_${m}_b32.log; exit 1; }
grep "inference:" $O/eval_${m}_b32.log
done
timeout -k 10 200 python -u tools/bench_eval.py deeplabv3plus_resnet101 1 > $O/eval_r101_b1.log 2>&1 || exit 1
grep "inference:" $O/eval_r101_b1.log
