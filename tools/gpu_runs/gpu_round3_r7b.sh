#!/bin/bash
set -u
O=gpurun_out/r7b; mkdir -p $O
timeout -k 10 500 python -u -m pytest tests/test_gpu_inference.py tests/test_gpu_sepconv.py tests/test_xception_pad.py -x -q --timeout 180 --timeout-method thread > $O/pytest.log 2>&1
rc=$?; tail -12 $O/pytest.log | cut -c1-300; [ $rc -ne 0 ] && exit $rc
for b in 1 16; do
timeout -k 10 200 python -u tools/bench_eval.py deeplabv3plus_xception65 $b > $O/eval_x65_b$b.log 2>&1 || { tail -5 $O/eval_x65_b$b.log; exit 1; }
grep "inference:" $O/eval_x65_b$b.log
done
timeout -k 10 240 python -u predict.py --model deeplabv3plus_xception65 --synthetic 128 --batch 8 --no-write > $O/predict_semseg_x65_b8.log 2>&1 || exit 1
tail -1 $O/predict_semseg_x65_b8.log | cut -c1-250
