# the reference's own workload (DANet-R101 instance task, trainBatch 16 over 4 GPUs = 4 per GPU, or 16 on one): eager vs graph-step
mkdir -p gpurun_out/r6o
for b in 4 16; do for m in "" "--graph-step"; do
  tag=$([ -z "$m" ] && echo eager || echo graph)
  timeout -k 10 300 python bench.py --model danet_resnet101 --task instance --crop 512 --batch $b --steps 24 --warmup 8 $m > gpurun_out/r6o/danet_b${b}_$tag.log 2>&1 || { tail -5 gpurun_out/r6o/danet_b${b}_$tag.log; exit 1; }
done; done
grep -o '"value": [0-9.]*' gpurun_out/r6o/*.log
