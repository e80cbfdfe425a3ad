#!/usr/bin/env python
"""Training entrypoint — same name and hyper-parameter names as `/root/reference/train_pascal.py`,
rebuilt as one process per GPU (the reference's TODO header `:1-8`: single ``train``/``val``
functions, distributed sampler, save on master only, non-blocking label copies, ``build_model``,
device from the local rank, explicit rendezvous).

    # 1 GPU, synthetic VOC-shaped data, the BASELINE.json headline config
    python train_pascal.py --synthetic --model deeplabv3plus_resnet101 --crop 513 --trainBatch 32
    # 8 GPUs on one node (RCCL over xGMI)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 \
        train_pascal.py --synthetic --model deeplabv3plus_resnet101 --sync-bn
    # CPU / gloo plumbing configuration
    python -m torch.distributed.run --nproc-per-node 2 --master-addr 127.0.0.1 train_pascal.py \
        --synthetic --device cpu --model fcn_resnet18 --crop 256 --nEpochs 1

Tasks
-----
``--task semseg`` (default): 21-class semantic segmentation with 255 = ignore, mIoU validation
(north star).  ``--task instance``: the reference's own workload — DEXTR-style binary instance
segmentation with a 4-channel input (RGB + n-ellipse guidance map), DANet, 512² object crops,
Jaccard at thresholds .3/.5/.8 (`train_pascal.py:123-145`, `:280-306`).

Defaults for ``nEpochs, trainBatch, testBatch, useTest, nTestInterval, snapshot, wd, momentum,
resume_epoch`` are the reference's (`:51-70`); ``lr`` defaults to the reference's 5e-8 only for the
instance task (it fine-tunes a converged checkpoint) and to DeepLab's 0.007 for semseg.
No telemetry of any kind is sent anywhere (the reference's hosted-experiment client and its
hard-coded key, `:10`, `:41`, are intentionally absent).
"""
from __future__ import annotations

import argparse
import os
import socket
import sys
import timeit
from collections import OrderedDict
from datetime import datetime

import numpy as np
import torch
from torch.utils.data import DataLoader

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

from distributedpytorch_amd import ops  # noqa: E402
from distributedpytorch_amd.data import CombineDBs, SyntheticVOC, VOCSegmentation, helpers  # noqa: E402
from distributedpytorch_amd.data import transforms as tr  # noqa: E402
from distributedpytorch_amd.models import build_model  # noqa: E402
from distributedpytorch_amd.parallel import (DistributedSampler, TrainEngine, dist_env,  # noqa: E402
                                             init_distributed)
from distributedpytorch_amd.utils import (JsonlLogger, LR_Scheduler, SegmentationMultiLosses,  # noqa: E402
                                          load_checkpoint, load_weights, save_checkpoint)
from distributedpytorch_amd.utils.checkpoint import run_dir  # noqa: E402


def parse_args(argv=None):
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    # ---- names and defaults of the reference's constants (train_pascal.py:51-70) ----
    ap.add_argument("--nEpochs", type=int, default=100)
    ap.add_argument("--resume_epoch", type=int, default=0)
    ap.add_argument("--trainBatch", type=int, default=16, help="per-process (per-GPU) batch")
    ap.add_argument("--testBatch", type=int, default=1)
    ap.add_argument("--useTest", type=int, default=1)
    ap.add_argument("--nTestInterval", type=int, default=1)
    ap.add_argument("--snapshot", type=int, default=100)
    ap.add_argument("--nInputChannels", type=int, default=None, help="4 for the instance task, else 3")
    ap.add_argument("--zero_pad_crop", type=int, default=1)
    ap.add_argument("--nAveGrad", type=int, default=1, help="gradient accumulation steps")
    ap.add_argument("--lr", type=float, default=None)
    ap.add_argument("--wd", type=float, default=5e-4)
    ap.add_argument("--momentum", type=float, default=0.9)
    ap.add_argument("--use_sbd", type=int, default=0)
    # ---- new knobs ----
    ap.add_argument("--task", choices=["semseg", "instance"], default="semseg")
    ap.add_argument("--model", default=None, help="see distributedpytorch_amd.models.MODEL_NAMES")
    ap.add_argument("--output-stride", type=int, default=None)
    ap.add_argument("--crop", type=int, default=None, help="513 (semseg) / 512 (instance)")
    ap.add_argument("--sync-bn", action="store_true")
    ap.add_argument("--freeze-bn", action="store_true")
    ap.add_argument("--head-lr-mult", type=float, default=1.0, help="10 = the 1x/10x groups of train_pascal.py:90-91")
    ap.add_argument("--lr-mode", choices=["poly", "cos", "step", "const"], default=None)
    ap.add_argument("--synthetic", action="store_true", help="VOC-shaped random data (no dataset on disk)")
    ap.add_argument("--synthetic-len", type=int, default=256)
    ap.add_argument("--data-root", default=None)
    ap.add_argument("--workers", type=int, default=2)
    ap.add_argument("--device", choices=["auto", "cpu"], default="auto")
    ap.add_argument("--impl", choices=["native", "stock"], default="native")
    ap.add_argument("--bucket-mb", type=int, default=64)
    ap.add_argument("--save-dir", default=None, help="default: run_<N> beside this script (as the reference)")
    ap.add_argument("--weights", default=None, help="state_dict to start from (module.-prefixed or not)")
    ap.add_argument("--max-iters", type=int, default=0, help="stop each epoch after N iterations (smoke runs)")
    ap.add_argument("--log-every", type=int, default=20)
    ap.add_argument("--seed", type=int, default=0)
    return ap.parse_args(argv)


# --------------------------------------------------------------------------------------
# data
# --------------------------------------------------------------------------------------
def build_transforms(args, train: bool):
    if args.task == "instance":
        size = (args.crop, args.crop)
        if train:  # train_pascal.py:123-134
            return tr.Compose([
                tr.RandomHorizontalFlip(),
                tr.ScaleNRotate(rots=(-20, 20), scales=(.75, 1.25)),
                tr.CropFromMaskStatic(crop_elems=("image", "gt"), relax=50, zero_pad=bool(args.zero_pad_crop)),
                tr.FixedResize(resolutions={"crop_image": size, "crop_gt": size}),
                tr.NEllipseWithGaussians(alpha=0.6, is_val=False),
                tr.ConcatInputs(elems=("crop_image", "nellipseWithGaussians")),
                tr.ToTensor()])
        return tr.Compose([  # train_pascal.py:135-145
            tr.CropFromMaskStatic(crop_elems=("image", "gt"), relax=50, zero_pad=bool(args.zero_pad_crop)),
            tr.FixedResize(resolutions={"void_pixels": None, "gt": None, "crop_image": size, "crop_gt": size}),
            tr.NEllipseWithGaussians(alpha=0.6, is_val=True),
            tr.ConcatInputs(elems=("crop_image", "nellipseWithGaussians")),
            tr.ToTensor()])
    if train:
        return tr.Compose([tr.RandomHorizontalFlip(), tr.RandomScaleCrop(args.crop, (0.5, 2.0)),
                           tr.Normalize(), tr.SemSegToTensor()])
    return tr.Compose([tr.FixedScaleCenterCrop(args.crop), tr.Normalize(), tr.SemSegToTensor()])


def build_datasets(args):
    t_tr, t_ts = build_transforms(args, True), build_transforms(args, False)
    if args.synthetic:
        mode = "instance" if args.task == "instance" else "semantic"
        raw = (args.crop, args.crop)
        db_train = SyntheticVOC(args.synthetic_len, raw, t_tr, mode, seed=args.seed)
        db_val = SyntheticVOC(max(args.synthetic_len // 8, 8), raw, t_ts, mode, seed=args.seed + 1)
        return db_train, db_val
    root = args.data_root or os.environ.get("VOC_ROOT")
    if root is None:
        raise SystemExit("give --data-root (or VOC_ROOT) or use --synthetic")
    semantic = args.task == "semseg"
    voc_train = VOCSegmentation(root=root, split="train", transform=t_tr, semantic=semantic, check_integrity=False)
    voc_val = VOCSegmentation(root=root, split="val", transform=t_ts, semantic=semantic, check_integrity=False)
    db_train = voc_train
    if args.use_sbd:  # train_pascal.py:150-152 (the reference's SBD import is commented out)
        raise SystemExit("SBD is not bundled; CombineDBs([voc_train, sbd], excluded=[voc_val]) is the hook")
    _ = CombineDBs  # exported for that hook
    return db_train, voc_val


def _collate_keys(task):
    return ("concat", "crop_gt") if task == "instance" else ("image", "gt")


# --------------------------------------------------------------------------------------
# train / validate (the reference's TODO "single fn train / single fn val")
# --------------------------------------------------------------------------------------
def train_epoch(engine, loader, sampler, scheduler, epoch, args, log):
    sampler.set_epoch(epoch)
    kin, kgt = _collate_keys(args.task)
    running = torch.zeros((), device=engine.device)
    n_iter, n_img = 0, 0
    t0 = timeit.default_timer()
    data_wait, t_fetch = 0.0, timeit.default_timer()
    for ii, batch in enumerate(loader):
        data_wait += timeit.default_timer() - t_fetch          # host time blocked on the loader workers
        if args.max_iters and ii >= args.max_iters:
            break
        scheduler(engine.optimizer, ii, epoch)
        inputs, gts = batch[kin], batch[kgt]
        loss = engine.train_step(inputs, gts)
        running += loss.float() * engine.accum_steps
        n_iter += 1
        n_img += inputs.shape[0]
        if args.log_every and (ii + 1) % args.log_every == 0:
            dist_env.check_finite_or_abort(loss, engine.global_step)   # one host sync per log interval
            dt = timeit.default_timer() - t0
            extra = {}
            if engine.on_gpu:
                extra["hbm_peak_gib"] = torch.cuda.max_memory_allocated(engine.device) / 2 ** 30
            log.log("train", epoch=epoch, iter=ii + 1, loss=float(running) / n_iter,
                    lr=engine.optimizer.param_groups[0]["lr"],
                    img_per_s=n_img * dist_env.get_world_size() / dt,
                    data_wait_ms=1e3 * data_wait / n_iter, **extra, **engine.timer.summary())
        t_fetch = timeit.default_timer()
    mean_loss = dist_env.all_reduce_scalar(float(running) / max(n_iter, 1), "sum", engine.device
                                           if engine.on_gpu else None) / dist_env.get_world_size()
    dt = timeit.default_timer() - t0
    return mean_loss, n_img * dist_env.get_world_size() / max(dt, 1e-9)


@torch.no_grad()
def validate_instance(engine, loader, args):
    """Per-object Jaccard at thresholds .3/.5/.8 with void pixels excluded (train_pascal.py:280-306)."""
    engine.model.eval()
    thresh = [0.3, 0.5, 0.8]
    jac = np.zeros(3)
    loss_sum, n = 0.0, 0
    relax = 50
    for ii, batch in enumerate(loader):
        if args.max_iters and ii >= args.max_iters:
            break
        x = engine.prepare_input(batch["concat"])
        out = engine.model(x)
        loss_sum += float(engine.criterion(out, batch["crop_gt"].to(engine.device)))
        pred = out[0] if isinstance(out, (tuple, list)) else out
        for jj in range(pred.shape[0]):
            p = torch.sigmoid(pred[jj, 0].float()).cpu().numpy()
            gt = helpers.tens2image(batch["gt"][jj])
            void = np.squeeze(helpers.tens2image(batch["void_pixels"][jj]))
            bbox = helpers.get_bbox(gt, pad=relax, zero_pad=bool(args.zero_pad_crop))
            if bbox is None:
                continue
            full = helpers.crop2fullmask(p, bbox, gt, zero_pad=bool(args.zero_pad_crop), relax=relax)
            for k, th in enumerate(thresh):
                jac[k] += helpers.calc_jaccard(full > th, gt, void)
            n += 1
    dev = engine.device if engine.on_gpu else None
    tot = [dist_env.all_reduce_scalar(v, "sum", dev) for v in (*jac.tolist(), loss_sum, float(n))]
    cnt = max(tot[4], 1.0)
    return {"jaccard": [t / cnt for t in tot[:3]], "loss": tot[3] / cnt}


class _KeyLoader:
    """Adapts dict batches to the (images, target) pairs ``TrainEngine.evaluate`` consumes."""

    def __init__(self, loader, kin, kgt, limit=0):
        self.loader, self.kin, self.kgt, self.limit = loader, kin, kgt, limit

    def __iter__(self):
        for i, b in enumerate(self.loader):
            if self.limit and i >= self.limit:
                return
            yield b[self.kin], b[self.kgt]


def main(argv=None):
    args = parse_args(argv)
    instance = args.task == "instance"
    args.crop = args.crop or (512 if instance else 513)
    args.lr = args.lr if args.lr is not None else (5e-8 if instance else 0.007)
    args.lr_mode = args.lr_mode or ("const" if instance else "poly")
    args.model = args.model or ("danet_resnet101" if instance else "deeplabv3plus_resnet101")
    args.nInputChannels = args.nInputChannels or (4 if instance else 3)
    info = init_distributed(device="cpu" if args.device == "cpu" else None)
    if args.impl == "stock":
        ops.set_backend("torch")
    if info.is_master and info.device.type == "cuda":
        print("Using GPU: {} ".format(info.device.index))
    torch.manual_seed(args.seed)
    np.random.seed(args.seed * 1000 + info.rank)

    save_root = args.save_dir or os.path.dirname(os.path.abspath(__file__))
    save_dir = run_dir(save_root, args.resume_epoch)
    exp_name = os.path.basename(os.path.abspath(save_root))
    model_name = args.model
    log = JsonlLogger(os.path.join(save_dir, "log.jsonl"))

    nclass = 1 if instance else 21
    net = build_model(args.model, nclass, output_stride=args.output_stride, sync_bn=args.sync_bn,
                      freeze_bn=args.freeze_bn, in_channels=args.nInputChannels)
    if info.is_master:
        print("No of parameters: ", sum(p.numel() for p in net.parameters() if p.requires_grad))
    criterion = SegmentationMultiLosses(nclass) if instance else ops.SoftmaxCrossEntropy(255)
    engine = TrainEngine(net, criterion, lr=args.lr, momentum=args.momentum, weight_decay=args.wd,
                         head_lr_mult=args.head_lr_mult, device=info.device, bucket_mb=args.bucket_mb,
                         accum_steps=args.nAveGrad)
    if args.weights:
        load_weights(args.weights, engine.ddp, strict=False)
    if args.resume_epoch:
        ck = os.path.join(save_dir, "models", model_name + "_epoch-" + str(args.resume_epoch - 1) + ".pth")
        print("Initializing weights from: {}".format(ck))
        load_checkpoint(ck, engine.ddp, engine.optimizer)

    db_train, db_val = build_datasets(args)
    sampler = DistributedSampler(len(db_train), shuffle=True, seed=args.seed, drop_last=True)
    val_sampler = DistributedSampler(len(db_val), shuffle=False, drop_last=False)
    pin = info.device.type == "cuda"
    trainloader = DataLoader(db_train, batch_size=args.trainBatch, sampler=sampler, num_workers=args.workers,
                             drop_last=True, pin_memory=pin, persistent_workers=args.workers > 0)
    test_bs = args.testBatch if instance else max(args.testBatch, 1)
    testloader = DataLoader(db_val, batch_size=test_bs, sampler=val_sampler, num_workers=args.workers,
                            pin_memory=pin)
    scheduler = LR_Scheduler(args.lr_mode, args.lr, args.nEpochs, len(trainloader))

    # parameter report (train_pascal.py:57-70, :156-159, :169)
    p = OrderedDict()
    p["trainBatch"], p["nAveGrad"], p["lr"], p["wd"], p["momentum"] = \
        args.trainBatch, args.nAveGrad, args.lr, args.wd, args.momentum
    p["world_size"], p["model"], p["task"], p["crop"] = info.world_size, args.model, args.task, args.crop
    p["optimizer"] = str(engine.optimizer)
    p["dataset_train"], p["dataset_test"] = str(db_train), str(db_val)
    p["transformations_train"] = [str(t) for t in db_train.transform.transforms]
    p["transformations_test"] = [str(t) for t in db_val.transform.transforms]
    p["host"], p["started"] = socket.gethostname(), datetime.now().strftime("%b%d_%H-%M-%S")
    if info.is_master:
        helpers.generate_param_report(os.path.join(save_dir, exp_name + ".txt"), p)

    best_val = 0.0
    kin, kgt = _collate_keys(args.task)
    if info.is_master:
        print("Training Network")
    for epoch in range(args.resume_epoch, args.nEpochs):
        start_time = timeit.default_timer()
        if info.is_master:
            print("Current LR", engine.optimizer.param_groups[0]["lr"])
        tr_loss, ips = train_epoch(engine, trainloader, sampler, scheduler, epoch, args, log)
        log.log("epoch", epoch=epoch, numImages=len(trainloader) * args.trainBatch * info.world_size,
                loss=tr_loss, img_per_s=ips)
        if (epoch % args.snapshot) == args.snapshot - 1 and epoch != 0:
            save_checkpoint(os.path.join(save_dir, "models", model_name + "_epoch-" + str(epoch) + ".pth"),
                            engine.ddp, engine.optimizer, {"epoch": epoch, "best_val": best_val})
        if args.useTest and epoch % args.nTestInterval == (args.nTestInterval - 1):
            if instance:
                res = validate_instance(engine, testloader, args)
                score = max(res["jaccard"])
                log.log("val", epoch=epoch, loss=res["loss"], JACCARD=res["jaccard"])
            else:
                res = engine.evaluate(_KeyLoader(testloader, kin, kgt, args.max_iters), 21, 255)
                score = res["miou"]
                log.log("val", epoch=epoch, loss=res["loss"], mIoU=res["miou"], pixel_acc=res["pixel_acc"])
            if score > best_val:
                best_val = score
                save_checkpoint(os.path.join(save_dir, "models", model_name + "_best.pth"), engine.ddp,
                                engine.optimizer, {"epoch": epoch, "best_val": best_val})
            log.log("time", epoch=epoch, execution_time=timeit.default_timer() - start_time)
    save_checkpoint(os.path.join(save_dir, "models", model_name + "_last.pth"), engine.ddp, engine.optimizer,
                    {"epoch": args.nEpochs - 1, "best_val": best_val})
    log.close()
    dist_env.shutdown()
    return best_val


if __name__ == "__main__":
    main()
