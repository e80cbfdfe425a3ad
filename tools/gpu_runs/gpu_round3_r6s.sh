#!/bin/bash
# prediction entrypoint: GPU tests + serving rates
set -u
O=gpurun_out/r6s; mkdir -p $O
timeout -k 10 420 python -u -m pytest tests/test_gpu_inference.py -x -v --timeout 180 --timeout-method thread > $O/pytest.log 2>&1
rc=$?; tail -15 $O/pytest.log; [ $rc -ne 0 ] && exit $rc
for b in 1 8 32; do
  timeout -k 10 240 python -u predict.py --synthetic 128 --batch $b --no-write > $O/semseg_r101_b$b.log 2>&1 || { tail -5 $O/semseg_r101_b$b.log; exit 1; }
  tail -1 $O/semseg_r101_b$b.log
done
timeout -k 10 240 python -u predict.py --synthetic 128 --batch 8 --no-write --no-graph > $O/semseg_r101_b8_eager.log 2>&1 || { tail -5 $O/semseg_r101_b8_eager.log; exit 1; }
tail -1 $O/semseg_r101_b8_eager.log
timeout -k 10 240 python -u predict.py --synthetic 128 --batch 1 --no-write --no-graph > $O/semseg_r101_b1_eager.log 2>&1 || { tail -5 $O/semseg_r101_b1_eager.log; exit 1; }
tail -1 $O/semseg_r101_b1_eager.log
for b in 1 8; do
  timeout -k 10 240 python -u predict.py --task instance --synthetic 64 --batch $b --no-write > $O/instance_danet_b$b.log 2>&1 || { tail -5 $O/instance_danet_b$b.log; exit 1; }
  tail -1 $O/instance_danet_b$b.log
done
timeout -k 10 240 python -u predict.py --model deeplabv3plus_xception65 --synthetic 64 --batch 8 --no-write > $O/semseg_x65_b8.log 2>&1 || { tail -5 $O/semseg_x65_b8.log; exit 1; }
tail -1 $O/semseg_x65_b8.log
