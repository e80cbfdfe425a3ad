#!/bin/bash
# templated evaluation epilogue: inference tests again, same-box A/B of the training step against the previous library
set -u
O=gpurun_out/r6v; mkdir -p $O
timeout -k 10 500 python -u -m pytest tests/test_gpu_inference.py -x -q --timeout 180 --timeout-method thread > $O/pytest_inference.log 2>&1
rc=$?; tail -3 $O/pytest_inference.log | cut -c1-300; [ $rc -ne 0 ] && exit $rc
for i in 1 2 3; do
  DPAMD_HIP_LIB=$PWD/distributedpytorch_amd/csrc/build/libdpamd_prev.so timeout -k 10 300 python bench.py > $O/r101_prev_$i.log 2>&1 || { tail -5 $O/r101_prev_$i.log; exit 1; }
  timeout -k 10 300 python bench.py > $O/r101_new_$i.log 2>&1 || { tail -5 $O/r101_new_$i.log; exit 1; }
done
grep -o '"value": [0-9.]*' $O/r101_*.log
for i in 1 2; do
DPAMD_HIP_LIB=$PWD/distributedpytorch_amd/csrc/build/libdpamd_prev.so timeout -k 10 300 python bench.py --model deeplabv3plus_xception65 --output-stride 8 > $O/x65_prev_$i.log 2>&1 || { tail -5 $O/x65_prev_$i.log; exit 1; }
timeout -k 10 300 python bench.py --model deeplabv3plus_xception65 --output-stride 8 > $O/x65_new_$i.log 2>&1 || { tail -5 $O/x65_new_$i.log; exit 1; }
done
grep -o '"value": [0-9.]*' $O/x65_*.log
for b in 1 8 32; do
  timeout -k 10 200 python -u tools/bench_eval.py deeplabv3plus_resnet101 $b > $O/eval_r101_b$b.log 2>&1 || { tail -5 $O/eval_r101_b$b.log; exit 1; }
  grep "inference:" $O/eval_r101_b$b.log
done
