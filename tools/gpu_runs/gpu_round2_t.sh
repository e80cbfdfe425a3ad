#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/t
A="--task texture --model deeplabv3plus_resnet101 --crop 257 --batch 16 --iters 300 --train-images 192 --val-images 48 --no-dropout --impls native"
run() { name=$1; shift; timeout -k 10 300 "$@" 2>gpurun_out/t/$name.err | tee gpurun_out/t/$name.log | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{\"impl\"'):
        r = json.loads(l); print('$name', r['impl'], 'miou', r['val_miou'], 'val_loss', r['val_loss'], 'curve', r['loss_curve'][2::4])
"; }
run default_a python -u tools/parity_run.py $A &&
run default_b python -u tools/parity_run.py $A &&
run default_c python -u tools/parity_run.py $A &&
run noprol_a env DPAMD_BN_PROLOGUE=0 python -u tools/parity_run.py $A &&
run noprol_b env DPAMD_BN_PROLOGUE=0 python -u tools/parity_run.py $A &&
run noprol_c env DPAMD_BN_PROLOGUE=0 python -u tools/parity_run.py $A
