# do the GPU tools still run on the pruned tree?  small arguments each; reports the first that fails and goes on (no GPU
# step follows a timeout: the loop stops at rc 124/137)
mkdir -p gpurun_out/r5e
run() { name=$1; shift; timeout -k 10 150 "$@" > gpurun_out/r5e/$name.log 2>&1; rc=$?; if [ $rc -eq 0 ]; then echo "ok $name"; else echo "FAILED($rc) $name: $(grep -v Warning gpurun_out/r5e/$name.log | tail -3 | tr '\n' ' ' | cut -c1-300)"; fi; [ $rc -eq 124 -o $rc -eq 137 ] && exit 1; return 0; }
run ablate_conv python tools/ablate_conv.py --cin 256 --cout 256 --hw 33 --k 3 --dil 1 --batch 16 --cfgs 5,9
run bench_bn python tools/bench_bn.py
run bench_conv python tools/bench_conv.py --batch 8 --filter l3
run bench_conv_splits python tools/bench_conv.py --batch 8 --filter l3.conv1 --splits 0,1,4
run bench_dw python tools/bench_dw.py --batch 4
run bench_eval python tools/bench_eval.py
run bench_shapes python tools/bench_shapes.py --batch 8 256:1024:33:1 768:768:65:1
run check_cfg python tools/check_cfg.py --batch 8 --a 5 --b 9 256:1024:33:1
run count_small_ops python tools/count_small_ops.py
run diag_chunks python tools/diag_chunks.py
run diag_model_step python tools/diag_model_step.py
run diag_nondeterminism python tools/diag_nondeterminism.py
run diag_sepconv python tools/diag_sepconv.py
run diag_xblocks python tools/diag_xblocks.py
run one_conv python tools/one_conv.py --cin 1024 --cout 256 --hw 33 --k 1 --batch 8
run overlap_probe python tools/overlap_probe.py
run stamp_conv python tools/stamp_conv.py --batch 8 --cfgs 5,9 256:1024:33:1
run gloo_gpu_probe python tools/gloo_gpu_probe.py
run p2p_probe python tools/p2p_probe.py
