#!/usr/bin/env python
"""Prediction from the command line (distributedpytorch_amd.inference).

    # semantic segmentation: every image of a directory -> paletted PNG label maps of the images' own sizes
    python predict.py --model deeplabv3plus_resnet101 --weights run_0/models/x_best.pth --input imgs/ --output out/

    # the reference's task: an object from its four extreme points (x,y pairs; one group per image)
    python predict.py --task instance --model danet_resnet101 --weights w.pth --input cat.jpg \
        --points "34,120 310,95 180,20 200,230" --output out/

    # no files: N synthetic images, to measure the serving rate of a model on this GPU
    python predict.py --synthetic 256 --batch 16

The reference has no such script (its network only runs inside `train_pascal.py`'s validation loop,
`/root/reference/train_pascal.py:237-298`); checkpoints written by it load here as they do in training
(utils.checkpoint.load_weights: ``module.`` prefixes, bare or wrapped state dicts, ``weights_only`` reads).
Prints one JSON line: images, seconds, images/sec (``forward_img_per_s``: with ``--synthetic`` the second pass over
the images in memory, PNG writing included unless ``--no-write``; with files the one pass, decode and writing included)."""
from __future__ import annotations

import argparse
import glob
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

EXTS = (".jpg", ".jpeg", ".png", ".bmp")


def parse_args(argv=None):
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--task", choices=["semseg", "instance"], default="semseg")
    ap.add_argument("--model", default=None, help="default: deeplabv3plus_resnet101 (semseg) / danet_resnet101 (instance)")
    ap.add_argument("--weights", default=None, help="checkpoint (.pth); without it the weights are random (a rate run)")
    ap.add_argument("--num-classes", type=int, default=None)
    ap.add_argument("--output-stride", type=int, default=None)
    ap.add_argument("--input", default=None, help="an image, a directory of images or a glob pattern")
    ap.add_argument("--output", default="predictions")
    ap.add_argument("--points", default=None, help="instance task: 'x,y x,y x,y x,y' per image, groups separated by ';'")
    ap.add_argument("--points-file", default=None, help="instance task: JSON {file name: [[x,y] x 4]}")
    ap.add_argument("--synthetic", type=int, default=0, help="N random images instead of --input")
    ap.add_argument("--synthetic-size", default="375x500", help="HxW of the synthetic images")
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--crop", type=int, default=None, help="network input side: 513 (semseg) / 512 (instance)")
    ap.add_argument("--fit", choices=["letterbox", "stretch"], default="letterbox")
    ap.add_argument("--threshold", type=float, default=0.5, help="instance task: probability threshold of the mask")
    ap.add_argument("--no-graph", action="store_true", help="launch the forward eagerly instead of replaying a hipGraph")
    ap.add_argument("--cpu", action="store_true")
    ap.add_argument("--no-write", action="store_true", help="skip the PNG files (rate runs)")
    ap.add_argument("--gpus", type=int, default=1,
                    help="N > 1: one process per GPU (started here, or ranks of torchrun), the images dealt round-robin; "
                         "rank 0 prints the job's total rate")
    return ap.parse_args(argv)


def _list_inputs(spec):
    if os.path.isdir(spec):
        files = sorted(os.path.join(spec, f) for f in os.listdir(spec) if f.lower().endswith(EXTS))
    elif os.path.isfile(spec):
        files = [spec]
    else:
        files = sorted(f for f in glob.glob(spec) if f.lower().endswith(EXTS))
    if not files:
        raise SystemExit(f"no images under {spec!r}")
    return files


def _points_for(args, files):
    import numpy as np
    if args.points_file:
        with open(args.points_file) as f:
            table = json.load(f)
        try:
            pts = [table[os.path.basename(p)] for p in files]
        except KeyError as e:
            raise SystemExit(f"--points-file has no entry for {e}")
    elif args.points:
        groups = [g for g in args.points.split(";") if g.strip()]
        pts = [[[float(v) for v in xy.split(",")] for xy in g.split()] for g in groups]
        if len(pts) != len(files):
            raise SystemExit(f"{len(files)} images but {len(pts)} point groups")
    else:
        raise SystemExit("the instance task needs --points or --points-file (four extreme points per image)")
    arr = np.asarray(pts, dtype=np.float32)
    if arr.shape != (len(files), 4, 2):
        raise SystemExit(f"four (x, y) points per image expected, got an array of shape {arr.shape}")
    return arr


def main(argv=None):
    args = parse_args(argv)
    # one command, N GPUs: the parent starts the ranks BEFORE anything touches the GPU (parallel.launch)
    from distributedpytorch_amd.parallel.launch import maybe_self_launch
    status = maybe_self_launch(args.gpus, argv=argv, script=os.path.abspath(__file__), cpu=args.cpu)
    if status is not None:
        sys.exit(status)
    import numpy as np
    import torch
    from PIL import Image
    from distributedpytorch_amd.inference import load_predictor, voc_palette

    instance = args.task == "instance"
    name = args.model or ("danet_resnet101" if instance else "deeplabv3plus_resnet101")
    crop = args.crop or (512 if instance else 513)
    rank, world = 0, 1
    if args.gpus > 1:
        from distributedpytorch_amd.parallel import dist_env, init_distributed
        info = init_distributed(device="cpu" if args.cpu else None)
        if info.world_size != args.gpus:
            raise SystemExit(f"[predict] --gpus {args.gpus} but the job has WORLD_SIZE={info.world_size}")
        rank, world, device = info.rank, info.world_size, info.device
    else:
        device = torch.device("cpu") if args.cpu or not torch.cuda.is_available() else torch.device("cuda", 0)
    kw = dict(crop=crop, batch=args.batch, graph=not args.no_graph)
    if not instance:
        kw["fit"] = args.fit
    torch.manual_seed(0)
    pred = load_predictor(args.task, name, args.weights, args.num_classes, args.output_stride, device, **kw)
    if not args.weights and rank == 0:
        print("[predict] no --weights: random initialisation (the rate is real, the labels are noise)", file=sys.stderr)

    if args.synthetic:
        h, w = (int(v) for v in args.synthetic_size.lower().split("x"))
        rng = np.random.default_rng(0)
        images = [rng.integers(0, 256, (h, w, 3), dtype=np.uint8) for _ in range(args.synthetic)]
        names = [f"synthetic_{i:05d}" for i in range(args.synthetic)]
        points = None
        if instance:
            cx, cy = rng.uniform(.3 * w, .7 * w, args.synthetic), rng.uniform(.3 * h, .7 * h, args.synthetic)
            rx, ry = rng.uniform(.1 * w, .25 * w, args.synthetic), rng.uniform(.1 * h, .25 * h, args.synthetic)
            points = np.stack([np.stack([cx - rx, cy], 1), np.stack([cx + rx, cy], 1),
                               np.stack([cx, cy - ry], 1), np.stack([cx, cy + ry], 1)], axis=1).astype(np.float32)
    else:
        if not args.input:
            raise SystemExit("give --input (or --synthetic N)")
        files = _list_inputs(args.input)
        points = _points_for(args, files) if instance else None
        n_files = len(files)
        if world > 1:            # this rank's share, dealt before anything is decoded: every world-th file
            files, points = files[rank::world], (points[rank::world] if points is not None else None)
        images = None            # decoded a group at a time (below): a directory need not fit into memory
        names = [os.path.splitext(os.path.basename(f))[0] for f in files]

    if args.synthetic:
        n_total = len(images)
        if world > 1:
            images, names = images[rank::world], names[rank::world]
            points = points[rank::world] if points is not None else None
    else:
        n_total = n_files

    pal = voc_palette().reshape(-1).tolist()

    def write(nms, results):
        if args.no_write:
            return
        os.makedirs(args.output, exist_ok=True)
        for nm, r in zip(nms, results):
            if instance:
                Image.fromarray(((r > args.threshold) * 255).astype(np.uint8), mode="L").save(
                    os.path.join(args.output, nm + ".png"))
            else:
                im = Image.fromarray(r, mode="P")
                im.putpalette(pal)
                im.save(os.path.join(args.output, nm + ".png"))

    def run(emit=False):
        """One pass over this rank's images in groups of 16 batches: decode (files), predict, write (``emit``)."""
        group = max(16 * args.batch, 64)
        for g0 in range(0, len(names), group):
            sl = slice(g0, g0 + group)
            imgs = images[sl] if images is not None else [np.asarray(Image.open(f).convert("RGB"), dtype=np.uint8)
                                                          for f in files[sl]]
            res = pred.predict(imgs, points[sl]) if instance else pred.predict(imgs)
            if emit:
                write(names[sl], res)

    sync = torch.cuda.synchronize if device.type == "cuda" else (lambda: None)
    t0 = time.perf_counter()
    run(emit=not args.synthetic)         # first pass: tile choices, graph captures (files: decode and PNG writing inside)
    sync()
    t_first = time.perf_counter() - t0
    t_steady = None
    if args.synthetic:           # a second pass over the same images is the steady rate
        t0 = time.perf_counter()
        run(emit=True)
        sync()
        t_steady = time.perf_counter() - t0
    if world > 1:
        # the job's rate: all images over the slowest rank's time
        t_first = dist_env.all_reduce_scalar(t_first, "max", device if device.type == "cuda" else None)
        if t_steady is not None:
            t_steady = dist_env.all_reduce_scalar(t_steady, "max", device if device.type == "cuda" else None)
        dist_env.barrier()
        dist_env.shutdown()
        if rank != 0:
            return None
    rec = {"task": args.task, "model": name, "images": n_total, "n_gpus": world, "batch": args.batch, "crop": crop,
           "device": str(device), "graph": bool(not args.no_graph and device.type == "cuda"),
           "first_pass_s": round(t_first, 3),
           "forward_img_per_s": round(n_total / (t_steady if t_steady is not None else t_first), 2),
           "steady": t_steady is not None, "output": None if args.no_write else args.output}
    print(json.dumps(rec), flush=True)
    return rec


if __name__ == "__main__":
    main()
