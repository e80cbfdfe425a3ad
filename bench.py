#!/usr/bin/env python
"""Headline benchmark: training images/sec of DeepLabv3+ ResNet-101 at 513×513, bf16, synthetic
VOC-shaped data (21 classes + ignore 255), random-init weights — the metric/config named in
BASELINE.json.  One process per GPU; for N>1 launch with

    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Each timed step is the full training step: GPU augmentation (random scale/crop/flip + normalise)
of a device-resident uint8 batch → forward → fused ignore-index CE → backward with overlapped
bucketed gradient all-reduce → fused SGD.  Weak scaling: per-GPU batch is fixed.

``--impl stock`` measures the stock PyTorch-ROCm path (MIOpen conv / ATen BN) on the same model
code — the self-established baseline of BASELINE.md; ``--impl native`` (default) runs the
hand-written HIP kernels.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=int(os.environ.get("DPAMD_BENCH_BATCH", "128")),
                    help="per-GPU batch (128 x 513² holds 55 GiB of the 288 GB HBM3E)")
    ap.add_argument("--crop", type=int, default=513)
    ap.add_argument("--model", default="deeplabv3plus_resnet101")
    ap.add_argument("--output-stride", type=int, default=None)
    ap.add_argument("--impl", choices=["native", "stock"], default=os.environ.get("DPAMD_BENCH_IMPL", "native"))
    ap.add_argument("--sync-bn", action="store_true")
    ap.add_argument("--bucket-mb", type=int, default=64)
    ap.add_argument("--cpu", action="store_true", help="plumbing run on CPU/gloo (tiny shapes)")
    ap.add_argument("--phases", action="store_true", help="print per-phase HIP-event timings (rank 0)")
    return ap.parse_args()


def main():
    args = parse_args()
    # hang watchdog: DPAMD_HANG_DUMP_S=N dumps every thread's Python stack to stderr
    # and exits if the run is still going after N seconds — a stuck collective names its call site
    if os.environ.get("DPAMD_HANG_DUMP_S"):
        import faulthandler
        faulthandler.dump_traceback_later(int(os.environ["DPAMD_HANG_DUMP_S"]), exit=True)
    from distributedpytorch_amd import ops
    from distributedpytorch_amd.models import build_model
    from distributedpytorch_amd.ops import SoftmaxCrossEntropy, affine_augment, make_affine_params
    from distributedpytorch_amd.parallel import TrainEngine, dist_env, init_distributed

    info = init_distributed(device="cpu" if args.cpu else None)
    if info.world_size != args.gpus and info.is_master:
        print(f"[bench] note: --gpus {args.gpus} but WORLD_SIZE={info.world_size}; using WORLD_SIZE",
              file=sys.stderr)
    on_gpu = info.device.type == "cuda"
    if args.impl == "stock":
        ops.set_backend("torch")
    torch.manual_seed(1234)  # identical init on every rank; DDP broadcasts anyway
    model = build_model(args.model, 21, output_stride=args.output_stride, sync_bn=args.sync_bn)
    engine = TrainEngine(model, SoftmaxCrossEntropy(255), lr=0.007, momentum=0.9, weight_decay=5e-4,
                         device=info.device, bucket_mb=args.bucket_mb)
    B, S = args.batch, args.crop
    g = torch.Generator().manual_seed(100 + info.rank)
    raw_img = torch.randint(0, 256, (B, S, S, 3), dtype=torch.uint8, generator=g).to(info.device)
    # blocky synthetic label map with a band of ignore pixels
    coarse = torch.randint(0, 21, (B, 1, 9, 9), generator=g).float()
    lab = torch.nn.functional.interpolate(coarse, size=(S, S), mode="nearest")[:, 0].to(torch.uint8)
    lab[:, : S // 32] = 255
    raw_lab = lab.to(info.device)
    rng = np.random.default_rng(7 + info.rank)

    def one_step():
        mats = make_affine_params(B, (S, S), (S, S), scales=(0.5, 2.0), flip_p=0.5, rng=rng)
        with engine.timer.phase("augment"):
            x, y = affine_augment(raw_img, raw_lab, torch.from_numpy(mats).to(info.device, non_blocking=True),
                                  (S, S), out_dtype=engine.compute_dtype)
        return engine.train_step(x, y)

    def sync():
        dist_env.barrier()
        if on_gpu:
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        loss = one_step()
    sync()
    engine.timer.summary(reset=True)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = one_step()
    sync()
    elapsed = time.perf_counter() - t0
    elapsed = dist_env.all_reduce_scalar(elapsed, "max", info.device if on_gpu else None)
    final_loss = float(loss.item())
    world = info.world_size
    ips = world * B * args.steps / elapsed
    if info.is_master:
        if args.phases:
            print("[bench] phases(ms):", {k: round(v, 3) for k, v in engine.timer.summary().items()},
                  file=sys.stderr)
            if on_gpu:
                print(f"[bench] peak HBM {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB", file=sys.stderr)
        rec = {
            "metric": ("train images/sec (node) DeepLabv3+ R101 513x513"
                   if (args.model == "deeplabv3plus_resnet101" and args.crop == 513)
                   else f"train images/sec (node) {args.model} {args.crop}x{args.crop}"),
            "value": round(ips, 3), "unit": "images/sec", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(1e3 * elapsed / args.steps, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "bf16" if on_gpu else "fp32", "data": "synthetic",
            "config": {"model": args.model, "global_batch": world * B, "per_gpu_batch": B,
                       "seq_len": S, "crop": S, "num_classes": 21,
                       "parallelism": f"dp{world}", "impl": args.impl, "sync_bn": bool(args.sync_bn),
                       "optimizer": "sgd-momentum", "final_loss": round(final_loss, 4)},
        }
        if args.model == "deeplabv3plus_resnet101" and args.crop == 513 and args.impl == "native" and on_gpu:
            # the reference publishes nothing (vs_baseline stays null); the self-established baseline
            # is the stock PyTorch-ROCm op path on the same model code (profiles/r1_stock_b32_bench.log, one MI355X)
            rec["self_baseline"] = {"stock_pytorch_rocm_images_per_sec_per_gpu": 251.8,
                                    "ratio_per_gpu": round(ips / world / 251.8, 3)}
        print(json.dumps(rec), flush=True)
    dist_env.shutdown()


if __name__ == "__main__":
    main()
