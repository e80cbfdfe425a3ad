#!/usr/bin/env python
"""Convergence parity of the hand-written-kernel path against the stock PyTorch-ROCm path
(BASELINE.json: "val mIoU parity"): the same model, seed, data, schedule and GPU
augmentation, trained once per implementation on a learnable synthetic VOC-shaped task
(class-coloured blocks + noise, 21 classes + a 255 ignore band), then evaluated on held-out images.

    python tools/parity_run.py [--model deeplabv3plus_resnet101] [--iters 200] [--batch 8] [--crop 513]

Prints one JSON line per implementation and a final verdict line."""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def make_data(n, size, seed, device):
    palette = torch.randint(30, 226, (21, 3), generator=torch.Generator().manual_seed(99)).float()  # shared by splits
    g = torch.Generator().manual_seed(seed)
    coarse = torch.randint(0, 21, (n, 1, 6, 6), generator=g).float()
    lab = torch.nn.functional.interpolate(coarse, size=(size, size), mode="nearest")[:, 0].long()
    img = palette[lab] + 12.0 * torch.randn(n, size, size, 3, generator=g)
    lab = lab.to(torch.uint8)
    lab[:, : size // 40] = 255
    return img.clamp(0, 255).to(torch.uint8).to(device), lab.to(device)


def run(impl, args):
    from distributedpytorch_amd import ops
    from distributedpytorch_amd.models import build_model
    from distributedpytorch_amd.ops import SoftmaxCrossEntropy, affine_augment, make_affine_params
    from distributedpytorch_amd.parallel import TrainEngine
    from distributedpytorch_amd.utils.lr_scheduler import LR_Scheduler
    ops.set_backend("torch" if impl == "stock" else "auto")
    dev = torch.device("cuda")
    torch.manual_seed(args.seed)
    model = build_model(args.model, 21)
    engine = TrainEngine(model, SoftmaxCrossEntropy(255), lr=args.lr, momentum=0.9, weight_decay=1e-4, device=dev)
    S, B = args.crop, args.batch
    train_img, train_lab = make_data(args.train_images, S, 1, dev)
    val_img, val_lab = make_data(args.val_images, S, 2, dev)
    rng = np.random.default_rng(args.seed)
    sched = LR_Scheduler("poly", args.lr, 1, args.iters)
    losses = []
    t0 = time.perf_counter()
    for it in range(args.iters):
        sched(engine.optimizer, it, 0)
        idx = torch.from_numpy(rng.integers(0, args.train_images, B)).to(dev)
        mats = make_affine_params(B, (S, S), (S, S), scales=(0.75, 1.25), flip_p=0.5, rng=rng)
        x, y = affine_augment(train_img[idx], train_lab[idx], torch.from_numpy(mats).to(dev), (S, S),
                              out_dtype=engine.compute_dtype)
        loss = engine.train_step(x, y)
        if it % 10 == 9 or it == 0:
            losses.append(round(float(loss), 4))
    torch.cuda.synchronize()
    train_s = time.perf_counter() - t0
    ident = make_affine_params(1, (S, S), (S, S), scales=(1.0, 1.0), flip_p=0.0, rng=np.random.default_rng(0))

    def val_batches():
        for i in range(0, args.val_images, B):
            n = min(B, args.val_images - i)
            m = torch.from_numpy(np.repeat(ident, n, 0)).to(dev)
            yield affine_augment(val_img[i:i + n], val_lab[i:i + n], m, (S, S), out_dtype=engine.compute_dtype)
    res = engine.evaluate(val_batches(), 21, 255)
    rec = {"impl": impl, "model": args.model, "iters": args.iters, "batch": B, "crop": S,
           "loss_curve": losses, "val_miou": round(res["miou"], 4), "val_pixel_acc": round(res["pixel_acc"], 4),
           "val_loss": round(res["loss"], 4), "train_seconds": round(train_s, 1)}
    print(json.dumps(rec), flush=True)
    del engine, model
    torch.cuda.empty_cache()
    return rec


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="deeplabv3plus_resnet101")
    ap.add_argument("--iters", type=int, default=200)
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--crop", type=int, default=513)
    ap.add_argument("--lr", type=float, default=0.02)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--train-images", type=int, default=96)
    ap.add_argument("--val-images", type=int, default=32)
    ap.add_argument("--impls", default="native,stock")
    args = ap.parse_args()
    recs = [run(i, args) for i in args.impls.split(",")]
    if len(recs) == 2:
        d = abs(recs[0]["val_miou"] - recs[1]["val_miou"])
        print(json.dumps({"verdict": "mIoU parity" if d < 0.03 else "mIoU differs", "abs_diff": round(d, 4),
                          "native": recs[0]["val_miou"], "stock": recs[1]["val_miou"]}))


if __name__ == "__main__":
    main()
