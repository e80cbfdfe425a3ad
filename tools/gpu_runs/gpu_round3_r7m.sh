#!/bin/bash
set -u
O=gpurun_out/r7m; mkdir -p $O
timeout -k 10 400 python -u -m pytest tests/test_gpu_inference.py -x -q -k "instance or predict" --timeout 200 --timeout-method thread > $O/pytest.log 2>&1
rc=$?; tail -3 $O/pytest.log | cut -c1-250; [ $rc -ne 0 ] && { grep -n "Error\|assert" $O/pytest.log | head; exit $rc; }
for b in 1 8 32; do
  timeout -k 10 240 python -u predict.py --task instance --synthetic 128 --batch $b --no-write > $O/predict_instance_danet_b$b.log 2>&1 || { tail -5 $O/predict_instance_danet_b$b.log; exit 1; }
  tail -1 $O/predict_instance_danet_b$b.log | cut -c1-250
done
