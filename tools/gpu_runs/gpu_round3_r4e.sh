#!/bin/bash
# deterministic mode end to end: the R101 texture run twice with the same seed in one process under DPAMD_DETERMINISTIC=1,
# tile choices fixed beforehand (a 5-iteration run writes the table the two long runs read) — loss curves and val mIoU
# must agree to the last digit; then the same pair WITHOUT the deterministic mode for contrast
set -o pipefail
mkdir -p gpurun_out/r4e
export TMPDIR=/tmp
export DPAMD_CONV_TUNE_FILE=$GRAFT_REPO_ROOT/gpurun_out/r4e/tune.json
R="--task texture --model deeplabv3plus_resnet101 --crop 257 --batch 16 --train-images 192 --val-images 48 --impls native"
DPAMD_DETERMINISTIC=1 timeout -k 10 200 python -u tools/parity_run.py $R --iters 5 --seeds 0 > gpurun_out/r4e/tune_run.log 2>gpurun_out/r4e/tune_run.err &&
DPAMD_DETERMINISTIC=1 timeout -k 10 300 python -u tools/parity_run.py $R --iters 300 --seeds 0,0 2>gpurun_out/r4e/det.err > gpurun_out/r4e/r101_deterministic_twice.log &&
timeout -k 10 300 python -u tools/parity_run.py $R --iters 300 --seeds 0,0 2>gpurun_out/r4e/nondet.err > gpurun_out/r4e/r101_default_twice.log
rc=$?
for f in gpurun_out/r4e/r101_*.log; do tail -1 $f | cut -c1-300; done
exit $rc
