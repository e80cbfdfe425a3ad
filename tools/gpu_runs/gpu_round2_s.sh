#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/s
A="--task texture --model deeplabv3plus_resnet101 --crop 257 --batch 16 --iters 300 --train-images 192 --val-images 48"
run() { name=$1; shift; timeout -k 10 300 "$@" 2>gpurun_out/s/$name.err | tee gpurun_out/s/$name.log | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{\"impl\"'):
        r = json.loads(l); print('$name', r['impl'], 'miou', r['val_miou'], 'val_loss', r['val_loss'], 'curve', r['loss_curve'][2::4])
"; }
run nodrop_native python -u tools/parity_run.py $A --no-dropout --impls native &&
run nodrop_stock python -u tools/parity_run.py $A --no-dropout --impls stock &&
run nodrop_native_noprol env DPAMD_BN_PROLOGUE=0 python -u tools/parity_run.py $A --no-dropout --impls native &&
run nodrop_native_noaux env DPAMD_AUX_STREAM=0 python -u tools/parity_run.py $A --no-dropout --impls native &&
run native_seed1 python -u tools/parity_run.py $A --impls native --seed 1 &&
run stock_seed1 python -u tools/parity_run.py $A --impls stock --seed 1
