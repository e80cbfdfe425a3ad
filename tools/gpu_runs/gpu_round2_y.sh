#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/y
A="--task texture --model deeplabv3plus_xception65 --output-stride 8 --crop 257 --batch 16 --iters 700 --train-images 192 --val-images 96 --impls native"
run() { name=$1; shift; timeout -k 10 400 "$@" 2>gpurun_out/y/$name.err | tee gpurun_out/y/$name.log | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{\"impl\"'):
        r = json.loads(l); print('$name', r['impl'], 'miou', r['val_miou'], 'val_loss', r['val_loss'], 'final', r['loss_curve'][-1], 'sec', r['train_seconds'])
"; }
run default python -u tools/parity_run.py $A &&
run nofold env DPAMD_SEPCONV_FOLD=0 python -u tools/parity_run.py $A &&
run noprol env DPAMD_BN_PROLOGUE=0 python -u tools/parity_run.py $A &&
run nofusedrop env DPAMD_FUSE_DROPOUT=0 python -u tools/parity_run.py $A &&
run dwgen3 env DPAMD_DWCONV_GEN=3 python -u tools/parity_run.py $A &&
run nodrop_native python -u tools/parity_run.py $A --no-dropout &&
run nodrop_stock python -u tools/parity_run.py $A --no-dropout --impls stock
