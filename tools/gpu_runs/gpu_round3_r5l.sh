# failure paths on the GPU: a diverging run (learning rate 1e4) with --on-nonfinite abort and restore, one and two ranks
mkdir -p gpurun_out/r5l
B="--synthetic --model deeplabv3plus_resnet18 --crop 129 --trainBatch 8 --testBatch 8 --synthetic-len 128 --nEpochs 3 --workers 2 --log-every 0 --hang-timeout 100 --finite-every 4 --snapshot 1 --useTest 0"
run() { name=$1; want=$2; shift; shift; timeout -k 10 200 "$@" > gpurun_out/r5l/$name.log 2>&1; rc=$?; echo "$name rc=$rc (expected $want): $(grep -i 'non-finite\|restor\|FloatingPoint' gpurun_out/r5l/$name.log | tail -2 | tr '\n' ' ' | cut -c1-260)"; [ $rc -eq 124 -o $rc -eq 137 ] && exit 1; return 0; }
run abort_1rank nonzero python -u train_pascal.py $B --lr 1e4 --save-dir /tmp/f1
run restore_1rank any python -u train_pascal.py $B --lr 1e4 --on-nonfinite restore --save-dir /tmp/f2
run abort_2ranks nonzero env DPAMD_DIST_BACKEND=gloo python -u train_pascal.py $B --lr 1e4 --gpus 2 --save-dir /tmp/f3
run restore_2ranks any env DPAMD_DIST_BACKEND=gloo python -u train_pascal.py $B --lr 1e4 --on-nonfinite restore --gpus 2 --save-dir /tmp/f4
