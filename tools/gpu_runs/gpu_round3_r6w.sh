#!/bin/bash
# full verification of the tree with the evaluation epilogue (v52) + validation / prediction rates
set -u
O=gpurun_out/r6w; mkdir -p $O
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread > $O/pytest.log 2>&1
rc=$?; tail -3 $O/pytest.log | cut -c1-200; [ $rc -ne 0 ] && { grep -n "FAILED\|Error" $O/pytest.log | head; exit $rc; }
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1 || { tail -5 $O/smoke.log; exit 1; }
tail -1 $O/smoke.log
timeout -k 10 300 python bench.py > $O/r101_b256.log 2>&1 || { tail -5 $O/r101_b256.log; exit 1; }
timeout -k 10 300 python bench.py --model deeplabv3plus_xception65 --output-stride 8 > $O/x65_b128.log 2>&1 || { tail -5 $O/x65_b128.log; exit 1; }
timeout -k 10 300 python bench.py --task instance --model danet_resnet101 --crop 512 > $O/danet_b64.log 2>&1 || { tail -5 $O/danet_b64.log; exit 1; }
grep -o '"value": [0-9.]*\|"tiles_autotuned_in_run": [0-9]*' $O/r101_b256.log $O/x65_b128.log $O/danet_b64.log
C="--synthetic --model deeplabv3plus_resnet101 --crop 513 --trainBatch 16 --synthetic-len 640 --nEpochs 2 --useTest 1 --workers 8 --log-every 0 --hang-timeout 120 --max-iters 6"
run() { name=$1; shift; timeout -k 10 300 python -u train_pascal.py $C "$@" > $O/$name.log 2>&1 || { echo "FAILED $name"; grep -v Warning $O/$name.log | tail -10 | cut -c1-250; exit 1; }; echo "$name: $(grep 'val:' $O/$name.log | tail -1 | cut -c1-170)"; }
run val_b1_device_graph --testBatch 1 --graph-eval on --save-dir /tmp/k2
run val_b1_device_eager --testBatch 1 --graph-eval off --save-dir /tmp/k1
run val_b32_device --testBatch 32 --save-dir /tmp/k3
C="--task instance --synthetic --model danet_resnet101 --crop 512 --trainBatch 8 --synthetic-len 256 --nEpochs 2 --useTest 1 --workers 8 --log-every 0 --hang-timeout 120 --max-iters 6"
run val_instance_b1 --testBatch 1 --save-dir /tmp/k4
run val_instance_b16 --testBatch 16 --save-dir /tmp/k5
for b in 1 8 32; do
  timeout -k 10 240 python -u predict.py --synthetic 128 --batch $b --no-write > $O/predict_semseg_r101_b$b.log 2>&1 || { tail -5 $O/predict_semseg_r101_b$b.log; exit 1; }
  tail -1 $O/predict_semseg_r101_b$b.log | cut -c1-250
done
timeout -k 10 240 python -u predict.py --synthetic 128 --batch 1 --no-write --no-graph > $O/predict_semseg_r101_b1_eager.log 2>&1 || exit 1
tail -1 $O/predict_semseg_r101_b1_eager.log | cut -c1-250
timeout -k 10 240 python -u predict.py --model deeplabv3plus_xception65 --synthetic 128 --batch 8 --no-write > $O/predict_semseg_x65_b8.log 2>&1 || exit 1
tail -1 $O/predict_semseg_x65_b8.log | cut -c1-250
for b in 1 8; do
  timeout -k 10 240 python -u predict.py --task instance --synthetic 64 --batch $b --no-write > $O/predict_instance_danet_b$b.log 2>&1 || { tail -5 $O/predict_instance_danet_b$b.log; exit 1; }
  tail -1 $O/predict_instance_danet_b$b.log | cut -c1-250
done
