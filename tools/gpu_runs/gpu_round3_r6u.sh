#!/bin/bash
# fused evaluation epilogue: tests, same-box A/B of the training step against the previous library, forward rates
set -u
O=gpurun_out/r6u; mkdir -p $O
true

for b in ; do
  timeout -k 10 200 python -u tools/bench_eval.py deeplabv3plus_resnet101 $b > $O/eval_r101_b$b.log 2>&1 || { tail -5 $O/eval_r101_b$b.log; exit 1; }
  grep "inference:" $O/eval_r101_b$b.log
done
for i in 1 2; do
  DPAMD_HIP_LIB=$PWD/distributedpytorch_amd/csrc/build/libdpamd_prev.so timeout -k 10 300 python bench.py > $O/r101_prev_$i.log 2>&1 || { tail -5 $O/r101_prev_$i.log; exit 1; }
  timeout -k 10 300 python bench.py > $O/r101_new_$i.log 2>&1 || { tail -5 $O/r101_new_$i.log; exit 1; }
done
grep -o '"value": [0-9.]*' $O/r101_*.log
for b in 1 8 32; do
  timeout -k 10 240 python -u predict.py --synthetic 128 --batch $b --no-write > $O/semseg_r101_b$b.log 2>&1 || { tail -5 $O/semseg_r101_b$b.log; exit 1; }
  tail -1 $O/semseg_r101_b$b.log | cut -c1-250
done
DPAMD_HIP_LIB=$PWD/distributedpytorch_amd/csrc/build/libdpamd_prev.so timeout -k 10 300 python bench.py --model deeplabv3plus_xception65 --output-stride 8 > $O/x65_prev.log 2>&1 || { tail -5 $O/x65_prev.log; exit 1; }
timeout -k 10 300 python bench.py --model deeplabv3plus_xception65 --output-stride 8 > $O/x65_new.log 2>&1 || { tail -5 $O/x65_new.log; exit 1; }
grep -o '"value": [0-9.]*' $O/x65_*.log
for m in deeplabv3plus_xception65 danet_resnet101; do
  timeout -k 10 200 python -u tools/bench_eval.py $m 8 > $O/eval_${m}_b8.log 2>&1 || { tail -5 $O/eval_${m}_b8.log; exit 1; }
  grep "inference:" $O/eval_${m}_b8.log
done
for b in 1 8; do
  timeout -k 10 240 python -u predict.py --task instance --synthetic 64 --batch $b --no-write > $O/instance_danet_b$b.log 2>&1 || { tail -5 $O/instance_danet_b$b.log; exit 1; }
  tail -1 $O/instance_danet_b$b.log | cut -c1-250
done
