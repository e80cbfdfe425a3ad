# same-box A/B: weight-gradient stream at the lowest priority (raw hipStreamCreateWithPriority) vs a normal-priority pool stream
mkdir -p gpurun_out/r4t
for rep in 1 2; do for m in aux-low aux-normal; do
  timeout -k 10 300 python tools/ab_loader_sync.py $m --steps 20 --warmup 8 > gpurun_out/r4t/r101_b256_${m}_${rep}.log 2>&1 || { tail -5 gpurun_out/r4t/r101_b256_${m}_${rep}.log; exit 1; }
done; done
for m in aux-low aux-normal; do
  timeout -k 10 300 python tools/ab_loader_sync.py $m --model deeplabv3plus_xception --output-stride 8 --batch 128 --steps 12 --warmup 5 > gpurun_out/r4t/x65_b128_${m}.log 2>&1 || exit 1
  timeout -k 10 300 python tools/ab_loader_sync.py $m --model danet_resnet101 --task instance --crop 512 --batch 64 --steps 20 --warmup 6 > gpurun_out/r4t/danet_b64_${m}.log 2>&1 || exit 1
done
grep -o '"value": [0-9.]*' gpurun_out/r4t/*.log
