# model x shape matrix through bench.py (3 steps each): odd crops and batches, every model family, both tasks, both output
# strides, the stock path; stops at the first failure
mkdir -p gpurun_out/r5b
run() { name=$1; shift; timeout -k 10 250 python bench.py --steps 3 --warmup 2 "$@" > gpurun_out/r5b/$name.log 2>&1 || { echo "FAILED $name"; tail -15 gpurun_out/r5b/$name.log | cut -c1-220; exit 1; }; echo "ok $name $(grep -o '"value": [0-9.]*\|"final_loss": [0-9.a-z]*' gpurun_out/r5b/$name.log | tr '\n' ' ')"; }
run fcn18_b3_crop321 --model fcn_resnet18 --batch 3 --crop 321
run fcn50_b17_crop257 --model fcn_resnet50 --batch 17 --crop 257
run fcn101_b5_crop400 --model fcn_resnet101 --batch 5 --crop 400
run dl18_b1 --model deeplabv3plus_resnet18 --batch 1 --crop 513
run dl50_os8_b6 --model deeplabv3plus_resnet50 --output-stride 8 --batch 6 --crop 385
run dl101_b7_crop449 --model deeplabv3plus_resnet101 --batch 7 --crop 449
run dl101_4ch --model deeplabv3plus_resnet101 --batch 4 --crop 257 --in-channels 4
run x65_os16_b5 --model deeplabv3plus_xception65 --output-stride 16 --batch 5 --crop 353
run x65_os8_b3 --model deeplabv3plus_xception65 --output-stride 8 --batch 3 --crop 289
run danet50_b3 --model danet_resnet50 --task instance --batch 3 --crop 384
run danet101_b5_semseg --model danet_resnet101 --task semseg --batch 5 --crop 256
run dl101_syncbn_forced_b9 --model deeplabv3plus_resnet101 --batch 9 --crop 321 --sync-bn
MIOPEN_FIND_MODE=FAST run stock_dl50_b4 --model deeplabv3plus_resnet50 --batch 4 --crop 257 --impl stock
