#!/bin/bash
set -u
O=gpurun_out/r7q; mkdir -p $O
timeout -k 10 400 python -u -m pytest tests/test_gpu_inference.py -x -q --timeout 200 --timeout-method thread > $O/pytest.log 2>&1
rc=$?; tail -3 $O/pytest.log | cut -c1-250; [ $rc -ne 0 ] && { grep -n "Error\|assert" $O/pytest.log | head; exit $rc; }
timeout -k 10 240 python -u predict.py --synthetic 512 --batch 32 --no-write > $O/predict_semseg_r101_b32.log 2>&1 || { tail -5 $O/predict_semseg_r101_b32.log; exit 1; }
tail -1 $O/predict_semseg_r101_b32.log | cut -c1-250
timeout -k 10 240 python -u predict.py --synthetic 64 --batch 8 --output $O/pngs > $O/predict_semseg_r101_b8_write.log 2>&1 || { tail -5 $O/predict_semseg_r101_b8_write.log; exit 1; }
tail -1 $O/predict_semseg_r101_b8_write.log | cut -c1-250; ls $O/pngs | wc -l; rm -rf $O/pngs
timeout -k 10 240 python -u predict.py --task instance --synthetic 128 --batch 8 --no-write > $O/predict_instance_b8.log 2>&1 || exit 1
tail -1 $O/predict_instance_b8.log | cut -c1-250
