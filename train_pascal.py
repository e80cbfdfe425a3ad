#!/usr/bin/env python
"""Training entrypoint — same name and hyper-parameter names as `/root/reference/train_pascal.py`,
rebuilt as one process per GPU (the reference's TODO header `:1-8`: single ``train``/``val``
functions, distributed sampler, save on master only, non-blocking label copies, ``build_model``,
device from the local rank, explicit rendezvous).

    # 1 GPU, synthetic VOC-shaped data, the BASELINE.json headline config
    python train_pascal.py --synthetic --model deeplabv3plus_resnet101 --crop 513 --trainBatch 128
    # 8 GPUs on one node (RCCL over xGMI): ONE command, as the reference (`train_pascal.py:92`) — the parent
    # starts the 8 rank processes itself before touching the GPU (parallel.launch) ...
    python train_pascal.py --gpus 8 --synthetic --model deeplabv3plus_resnet101 --sync-bn
    # ... or under torchrun (HSA_ENABLE_IPC_MODE_LEGACY=0 must be exported: dmabuf IPC for RCCL / HIP IPC)
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
from distributedpytorch_amd.utils import sanity  # noqa: E402


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
    ap.add_argument("--gpus", type=int, default=1, help="N > 1: start N rank processes (one per GPU) from this command")
    ap.add_argument("--data-path", choices=["auto", "gpu", "cpu"], default="auto",
                    help="gpu: decoded uint8 dataset resident in HBM + one augmentation gather per step "
                         "(data.gpu_pipeline); cpu: the reference's worker-side transforms (behaviour parity)")
    ap.add_argument("--val-path", choices=["auto", "cpu"], default="auto",
                    help="auto: validation follows --data-path (device store + device metric); cpu: the reference's host "
                         "validation (DataLoader transforms, per-object NumPy Jaccard) even on the GPU data path")
    ap.add_argument("--gpu-cache-gb", type=float, default=64.0, help="HBM budget of the device sample store; larger "
                    "datasets stream raw uint8 batches one step ahead instead")
    ap.add_argument("--sbd-root", default=None, help="SBD benchmark_RELEASE root for --use_sbd 1")
    ap.add_argument("--hang-timeout", type=int, default=int(os.environ.get("DPAMD_HANG_DUMP_S", "0")),
                    help="seconds without a finished log interval before every thread's stack is dumped and the "
                         "rank exits (0 = off)")
    ap.add_argument("--graph-step", action="store_true",
                    help="replay the training step from one hipGraph per input shape (one rank, no accumulation): the "
                         "small-batch regime, where the host's ~900 launches per step are the bound")
    ap.add_argument("--graph-eval", choices=["auto", "on", "off"], default="auto",
                    help="validation: replay the forward from a hipGraph per batch shape.  auto: with --graph-step, and "
                         "for the semantic task on the GPU when testBatch <= 4 (the reference's testBatch = 1 is "
                         "launch-bound: 234 -> 310 img/s)")
    ap.add_argument("--finite-every", type=int, default=50, help="steps between host reads of the device-side "
                    "non-finite-loss flag")
    ap.add_argument("--on-nonfinite", choices=["abort", "restore"], default="abort",
                    help="restore: reload the last checkpoint of this run and continue (at most 3 times)")
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


def build_datasets(args, raw: bool = False):
    """``raw``: datasets without transforms (the GPU data path decodes on the host and augments on the device)."""
    t_tr = None if raw else build_transforms(args, True)
    t_ts = build_transforms(args, False)
    if args.synthetic:
        mode = "instance" if args.task == "instance" else "semantic"
        size = (args.crop, args.crop)
        db_train = SyntheticVOC(args.synthetic_len, size, t_tr, mode, seed=args.seed)
        db_val = SyntheticVOC(max(args.synthetic_len // 8, 8), size, t_ts, mode, seed=args.seed + 1)
        return db_train, db_val
    root = args.data_root or os.environ.get("VOC_ROOT")
    if root is None:
        raise SystemExit("give --data-root (or VOC_ROOT) or use --synthetic")
    semantic = args.task == "semseg"
    voc_train = VOCSegmentation(root=root, split="train", transform=t_tr, semantic=semantic, check_integrity=False)
    voc_val = VOCSegmentation(root=root, split="val", transform=t_ts, semantic=semantic, check_integrity=False)
    db_train = voc_train
    if args.use_sbd:  # train_pascal.py:150-152: CombineDBs([voc_train, sbd], excluded=[voc_val])
        from distributedpytorch_amd.data.sbd import SBDSegmentation
        sbd_root = args.sbd_root or os.environ.get("SBD_ROOT")
        if sbd_root is None:
            raise SystemExit("--use_sbd 1 needs --sbd-root (or SBD_ROOT): the SBD benchmark_RELEASE directory")
        sbd = SBDSegmentation(root=sbd_root, split=["train", "val"], transform=t_tr, semantic=semantic)
        db_train = CombineDBs([voc_train, sbd], excluded=[voc_val])
    return db_train, voc_val


def _collate_keys(task):
    return ("concat", "crop_gt") if task == "instance" else ("image", "gt")


# --------------------------------------------------------------------------------------
# train / validate (the reference's TODO "single fn train / single fn val")
# --------------------------------------------------------------------------------------
class StepGuard:
    """Failure detection on the hot loop without host syncs: a device-side counter of
    non-finite losses (read every ``every`` steps) and a hang watchdog that is re-armed whenever the loop makes
    progress — if no step retires for ``hang_s`` seconds every thread's stack goes to stderr and the rank exits,
    so a stuck collective names its call site instead of hanging the job."""

    def __init__(self, device, every: int, hang_s: int):
        self.bad = torch.zeros((), dtype=torch.int32, device=device)
        self.every, self.hang_s, self.n = max(int(every), 1), int(hang_s), 0
        self.device = device
        if self.hang_s > 0:
            import faulthandler
            faulthandler.dump_traceback_later(self.hang_s, exit=True)

    def step(self, loss: torch.Tensor, global_step: int) -> None:
        self.bad += (~torch.isfinite(loss.detach())).to(torch.int32)      # device-side, no sync
        self.n += 1
        if self.n % self.every == 0:
            self.check(global_step)

    def check(self, global_step: int) -> None:
        bad = float(self.bad.item())                                       # the one host read per interval
        if self.hang_s > 0:
            import faulthandler
            faulthandler.dump_traceback_later(self.hang_s, exit=True)      # progress was made: re-arm
        dev = self.device if self.device.type == "cuda" else None
        if dist_env.all_reduce_scalar(bad, "max", dev) > 0:
            self.bad.zero_()
            raise FloatingPointError(f"non-finite loss within the {self.every} steps before step {global_step} "
                                     "on at least one rank")

    def progress(self) -> None:
        """Re-arm the watchdog without a host read: validation batches, checkpoint writes and figures are progress
        too — without this an evaluation longer than ``hang_s`` would be taken for a hang."""
        if self.hang_s > 0:
            import faulthandler
            faulthandler.dump_traceback_later(self.hang_s, exit=True)

    def close(self) -> None:
        if self.hang_s > 0:
            import faulthandler
            faulthandler.cancel_dump_traceback_later()


def _graph_eval(args) -> bool:
    mode = getattr(args, "graph_eval", "auto")               # (callers outside main() pass their own namespace)
    if mode != "auto":
        return mode == "on"
    if getattr(args, "graph_step", False):
        return True
    # semantic task: the batch-1 forward is still bound by its launches (R101: 2.9 ms eager, 2.3 ms replayed); the
    # instance task's DANet forward at batch 1 is bound by the device since the evaluation epilogue (145 img/s eager,
    # 129 replayed: profiles/r3_eval_forward_fused.md)
    return getattr(args, "task", "semseg") != "instance" and getattr(args, "testBatch", 1) <= 4


def _memory_report(engine) -> dict:
    """Host resident set of this process and the device allocator's state, per epoch: a leak shows as a slope."""
    out = {}
    try:
        import psutil
        out["host_rss_gib"] = psutil.Process().memory_info().rss / 2 ** 30
    except Exception:                                        # psutil is optional
        pass
    if engine.on_gpu:
        out["hbm_allocated_gib"] = torch.cuda.memory_allocated(engine.device) / 2 ** 30
        out["hbm_reserved_gib"] = torch.cuda.memory_reserved(engine.device) / 2 ** 30
    return out


def train_epoch(engine, loader, scheduler, epoch, args, log, guard=None, checker=None):
    """One epoch.  ``loader`` yields ``(inputs, target)`` — already augmented device tensors on the GPU data path,
    host tensors from the DataLoader workers on the CPU path."""
    loader.set_epoch(epoch)
    running = torch.zeros((), device=engine.device)
    n_iter, n_img = 0, 0
    t0 = timeit.default_timer()
    data_wait, t_fetch = 0.0, timeit.default_timer()
    # `data_wait` is host wall time inside the loader: real starvation on the CPU path, but on the device path (1 ms of
    # host work per batch, tools/loader_host_cost.py) mostly the launch queue pushing back on a host that is a step
    # ahead.  `busy` tells the two apart: the share of fetches that BEGAN with the step's stream still working.
    stream = torch.cuda.current_stream(engine.device) if engine.on_gpu else None
    busy = 0
    for ii, (inputs, gts) in enumerate(loader):
        data_wait += timeit.default_timer() - t_fetch          # host time blocked on the data path
        if args.max_iters and ii >= args.max_iters:
            break
        scheduler(engine.optimizer, ii, epoch)
        if checker is not None:                                # DPAMD_DEBUG: the reference's per-batch asserts
            checker.check(inputs, gts)
        loss = engine.train_step(inputs, gts)
        running += loss.float() * engine.accum_steps
        n_iter += 1
        n_img += inputs.shape[0]
        if guard is not None:
            guard.step(loss, engine.global_step)
        if args.log_every and (ii + 1) % args.log_every == 0:
            dt = timeit.default_timer() - t0
            extra = {}
            if engine.on_gpu:
                extra["hbm_peak_gib"] = torch.cuda.max_memory_allocated(engine.device) / 2 ** 30
                extra["gpu_busy_at_fetch"] = busy / max(ii, 1)
            log.log("train", epoch=epoch, iter=ii + 1, loss=float(running) / n_iter,
                    lr=engine.optimizer.param_groups[0]["lr"],
                    img_per_s=n_img * dist_env.get_world_size() / dt,
                    data_wait_ms=1e3 * data_wait / n_iter, **extra, **engine.timer.summary())
        if stream is not None and not stream.query():
            busy += 1
        t_fetch = timeit.default_timer()
    if guard is not None:
        guard.check(engine.global_step)
    mean_loss = dist_env.all_reduce_scalar(float(running) / max(n_iter, 1), "sum", engine.device
                                           if engine.on_gpu else None) / dist_env.get_world_size()
    if engine.on_gpu:
        torch.cuda.synchronize(engine.device)
    dt = timeit.default_timer() - t0
    return mean_loss, n_img * dist_env.get_world_size() / max(dt, 1e-9), 1e3 * data_wait / max(n_iter, 1)


class HostLoader:
    """The reference's data path: DataLoader workers run the CPU transforms; dict batches → (inputs, target)."""

    def __init__(self, loader, sampler, kin, kgt, generator=None):
        self.loader, self.sampler, self.kin, self.kgt, self.generator = loader, sampler, kin, kgt, generator
        self.epoch = 0

    def set_epoch(self, epoch):
        self.epoch = epoch
        self.sampler.set_epoch(epoch)

    def state_dict(self):
        """The generator the DataLoader draws its workers' base seeds from (this rank's own) and the sampler epoch;
        NumPy's global state travels with ``utils.checkpoint.capture_rng_state``."""
        st = {"kind": "host", "epoch": int(self.epoch)}
        if self.generator is not None:
            st["generator"] = self.generator.get_state()
        return st

    def load_state_dict(self, st):
        if not st or st.get("kind") != "host":
            return
        if self.generator is not None and "generator" in st:
            self.generator.set_state(st["generator"])
        self.set_epoch(int(st.get("epoch", 0)))

    def reseed(self, seed, rank, salt):
        if self.generator is not None:
            self.generator.manual_seed(seed * 1000 + rank + 7919 * salt)

    def __len__(self):
        return len(self.loader)

    def __iter__(self):
        for b in self.loader:
            yield b[self.kin], b[self.kgt]


@torch.no_grad()
def validate_instance(engine, loader, args, epoch=0, fig_path=None, tick=None):
    """Per-object Jaccard at thresholds .3/.5/.8 with void pixels excluded (train_pascal.py:280-306); the first
    batch's prediction and attention maps go to a 4-panel PNG (train_pascal.py:257-278)."""
    engine.model.eval()
    thresh = [0.3, 0.5, 0.8]
    jac = np.zeros(3)
    loss_sum, n = 0.0, 0
    relax = 50
    for ii, batch in enumerate(loader):
        if args.max_iters and ii >= args.max_iters:
            break
        if tick is not None:
            tick()
        x = engine.prepare_input(batch["concat"])
        out = engine.model(x)
        loss_sum += float(engine.criterion(out, batch["crop_gt"].to(engine.device)))
        outs = out if isinstance(out, (tuple, list)) else (out,)
        pred = outs[0]
        if ii == 0 and fig_path is not None and dist_env.is_master():
            from distributedpytorch_amd.utils.visualize import save_validation_panels
            save_validation_panels(fig_path, batch["concat"][0], batch["crop_gt"][0, 0], [o[0] for o in outs], epoch)
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


@torch.no_grad()
def validate_instance_device(engine, loader, args, epoch=0, fig_path=None, tick=None):
    """``validate_instance`` without leaving the device: the validation chain (mask bbox -> relax-50
    crop -> resize -> fixed extreme points -> guidance -> concat) runs in ``GpuBatchLoader(train=False)``, the
    predictions are pasted back, thresholded and scored by ``ops.paste_jaccard`` against the full-size masks in the
    device store, and the host reads ONE vector per epoch (the CPU path copies every prediction to the host and
    scores one object at a time in NumPy)."""
    from distributedpytorch_amd.ops.augment import paste_jaccard
    engine.model.eval()
    thresh = (0.3, 0.5, 0.8)
    dev = engine.device
    acc = torch.zeros(5, dtype=torch.float64, device=dev)            # Σ jaccard (3 thresholds), Σ loss, #objects
    t0, n_img = timeit.default_timer(), 0
    fwd = engine.inference_forward(_graph_eval(args))                # testBatch <= 4: replayed from a hipGraph
    for ii, (x, y, meta) in enumerate(loader.iter_eval()):
        if args.max_iters and ii >= args.max_iters:
            break
        if tick is not None:
            tick()
        out = fwd(engine.prepare_input(x))
        outs = out if isinstance(out, (tuple, list)) else (out,)
        bb = meta["crop_bbox"]
        has = (bb[:, 2] >= bb[:, 0]).double()                        # an empty mask has no box: skipped, as on the host
        cnt = paste_jaccard(outs[0], bb, meta["gt"], meta["void"], meta["src_idx"], meta["sizes"], 50, thresh,
                            validated=True).double()
        jac = torch.where(cnt[..., 1] > 0, cnt[..., 0] / cnt[..., 1].clamp_min(1.0), torch.ones_like(cnt[..., 0]))
        acc[:3] += (jac * has[:, None]).sum(0)
        acc[3] += engine.criterion(out, y).double() * x.shape[0]
        acc[4] += has.sum()
        n_img += x.shape[0]
        if ii == 0 and fig_path is not None and dist_env.is_master():
            from distributedpytorch_amd.utils.visualize import save_validation_panels
            save_validation_panels(fig_path, x[0].float().cpu(), y[0, 0].float().cpu(),
                                   [o[0].float().cpu() for o in outs], epoch)
    if dist_env.get_world_size() > 1:
        dist_env.all_reduce_(acc)            # (host-staged under gloo: its device-tensor path stalls with shared GPUs)
    tot = acc.tolist()                                               # the epoch's one host read
    cnt = max(tot[4], 1.0)
    dt = max(timeit.default_timer() - t0, 1e-9)
    return {"jaccard": [t / cnt for t in tot[:3]], "loss": tot[3] / cnt,
            "img_per_s": n_img * dist_env.get_world_size() / dt}


class _Ticked:
    """Device validation batches for ``TrainEngine.evaluate``: at most ``limit`` of them, the watchdog re-armed."""

    def __init__(self, loader, limit=0, tick=None):
        self.loader, self.limit, self.tick = loader, limit, tick

    def __iter__(self):
        for i, b in enumerate(self.loader):
            if self.limit and i >= self.limit:
                return
            if self.tick is not None:
                self.tick()
            yield b


class _KeyLoader:
    """Adapts dict batches to the (images, target) pairs ``TrainEngine.evaluate`` consumes."""

    def __init__(self, loader, kin, kgt, limit=0, tick=None):
        self.loader, self.kin, self.kgt, self.limit, self.tick = loader, kin, kgt, limit, tick

    def __iter__(self):
        for i, b in enumerate(self.loader):
            if self.limit and i >= self.limit:
                return
            if self.tick is not None:
                self.tick()                       # StepGuard.progress: a validation batch retired
            yield b[self.kin], b[self.kgt]


def _seed_worker(base_seed):
    """DataLoader ``worker_init_fn``: NumPy's global generator (every CPU transform draws from it) seeded from
    (seed, rank, worker id, torch's per-epoch worker seed) — without it worker w of EVERY rank replays the same
    flips / scales / rotations (the reference has one process, so it never met this)."""
    def init(worker_id):
        info = torch.utils.data.get_worker_info()
        ss = np.random.SeedSequence([base_seed, worker_id, int(info.seed) % (2 ** 32)])
        np.random.seed(ss.generate_state(4))
        import random
        random.seed(int(ss.generate_state(1)[0]))
    return init


def _semseg_figure(engine, loader, fig_path, epoch):
    """4-panel PNG of the first validation sample for the semantic task (prediction = arg-max class map)."""
    from distributedpytorch_amd.utils.visualize import save_validation_panels
    engine.model.eval()
    with torch.no_grad():
        for images, target in loader:
            out = engine.model(engine.prepare_input(images[:1]))
            outs = out if isinstance(out, (tuple, list)) else (out,)
            save_validation_panels(fig_path, images[0], target[0], [o[0] for o in outs], epoch, semantic=True)
            return


def main(argv=None):
    args = parse_args(argv)
    # ---- one command, N GPUs (the reference's `train_pascal.py:92`): launch the ranks before any GPU call ----
    from distributedpytorch_amd.parallel.launch import maybe_self_launch
    status = maybe_self_launch(args.gpus, argv=argv, script=os.path.abspath(__file__), cpu=args.device == "cpu")
    if status is not None:
        if status:
            sys.exit(status)
        return None
    instance = args.task == "instance"
    args.crop = args.crop or (512 if instance else 513)
    args.lr = args.lr if args.lr is not None else (5e-8 if instance else 0.007)
    args.lr_mode = args.lr_mode or ("const" if instance else "poly")
    args.model = args.model or ("danet_resnet101" if instance else "deeplabv3plus_resnet101")
    args.nInputChannels = args.nInputChannels or (4 if instance else 3)
    info = init_distributed(device="cpu" if args.device == "cpu" else None)
    if args.gpus > 1 and info.world_size != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but the job has WORLD_SIZE={info.world_size}")
    if args.impl == "stock":
        ops.set_backend("torch")
    on_gpu = info.device.type == "cuda"
    if info.is_master and on_gpu:
        print("Using GPU: {} ".format(info.device.index))
    torch.manual_seed(args.seed)                       # identical model init on every rank
    np.random.seed(args.seed * 1000 + info.rank)
    data_path = args.data_path
    if data_path == "auto":
        data_path = "gpu" if on_gpu else "cpu"

    save_root = args.save_dir or os.path.dirname(os.path.abspath(__file__))
    save_dir = run_dir(save_root, args.resume_epoch)
    exp_name = os.path.basename(os.path.abspath(save_root))
    model_name = args.model
    log = JsonlLogger(os.path.join(save_dir, "log.jsonl"))

    nclass = 1 if instance else 21
    net = build_model(args.model, nclass, output_stride=args.output_stride, sync_bn=args.sync_bn,
                      freeze_bn=args.freeze_bn, in_channels=args.nInputChannels)
    if info.is_master:
        n_params = sum(p.numel() for p in net.parameters() if p.requires_grad)
        backbone = getattr(net, "backbone", None)
        if hasattr(backbone, "logical_num_parameters"):     # Xception's 728-channel flow may be stored 768 wide
            n_params -= sum(p.numel() for p in backbone.parameters()) - backbone.logical_num_parameters()
        print("No of parameters: ", n_params)
        if on_gpu and not args.graph_step and args.trainBatch <= 16 and args.nAveGrad == 1 and not args.sync_bn:
            print("(per-GPU batch {}: the step is bound by the host's launches here; --graph-step replays it from one "
                  "hipGraph — R101 B16 513 -> 662 img/s, DANet B4 134 -> 217)".format(args.trainBatch))
    criterion = SegmentationMultiLosses(nclass) if instance else ops.SoftmaxCrossEntropy(255)
    engine = TrainEngine(net, criterion, lr=args.lr, momentum=args.momentum, weight_decay=args.wd,
                         head_lr_mult=args.head_lr_mult, device=info.device, bucket_mb=args.bucket_mb,
                         accum_steps=args.nAveGrad, graph_step=args.graph_step)
    if args.weights:
        load_weights(args.weights, engine.ddp, strict=False)
    best_val = 0.0
    last_ckpt = None
    resumed_rank_state = {}
    if args.resume_epoch:
        ck = os.path.join(save_dir, "models", model_name + "_epoch-" + str(args.resume_epoch - 1) + ".pth")
        if not os.path.isfile(ck):          # every rank checks the same path: all raise together, none is left waiting
            raise FileNotFoundError(f"--resume_epoch {args.resume_epoch}: {ck} does not exist")
        if info.is_master:
            print("Initializing weights from: {}".format(ck))
        extra = load_checkpoint(ck, engine.ddp, engine.optimizer)
        # the sidecar carries what the reference keeps in a variable across epochs (train_pascal.py:177, :301-304):
        # without it the first validation after a resume would overwrite <model>_best.pth with a worse model
        best_val = float(extra.get("best_val", 0.0))
        resumed_rank_state = extra.get("rank_state", {})
        last_ckpt = ck
        if info.is_master:
            print("Resumed: epoch {} best_val {:.4f}".format(extra.get("epoch", args.resume_epoch - 1), best_val))

    # ---- data ----
    kin, kgt = _collate_keys(args.task)
    pin = on_gpu
    if data_path == "gpu":
        from distributedpytorch_amd.data.gpu_pipeline import GpuBatchLoader
        db_train, db_val = build_datasets(args, raw=True)
        sampler = DistributedSampler(len(db_train), shuffle=True, seed=args.seed, drop_last=True)
        rots, scales = ((-20, 20), (.75, 1.25)) if instance else ((0.0, 0.0), (0.5, 2.0))
        trainloader = GpuBatchLoader(db_train, sampler, args.trainBatch, info.device, args.crop, args.task,
                                     scales=scales, rots=rots, seed=args.seed, rank=info.rank, workers=args.workers,
                                     hbm_budget_gb=args.gpu_cache_gb, relax=50, zero_pad=bool(args.zero_pad_crop))
        t_start = timeit.default_timer()
        trainloader.prepare()
        if info.is_master and trainloader.store is not None:
            print("Device sample store: {} samples, {:.2f} GiB, filled in {:.1f} s".format(
                len(db_train), trainloader.store.nbytes / 2 ** 30, timeit.default_timer() - t_start))
        train_tf = ["GPU: " + ("Flip+ScaleNRotate+CropFromMaskStatic+FixedResize+NEllipseWithGaussians+ConcatInputs"
                               if instance else "RandomScaleCrop+Flip+Normalize") + " (ops.augment)"]
    else:
        db_train, db_val = build_datasets(args)
        sampler = DistributedSampler(len(db_train), shuffle=True, seed=args.seed, drop_last=True)
        gen = torch.Generator().manual_seed(args.seed * 1000 + info.rank)     # rank-dependent worker base seeds
        loader = DataLoader(db_train, batch_size=args.trainBatch, sampler=sampler, num_workers=args.workers,
                            drop_last=True, pin_memory=pin, persistent_workers=args.workers > 0, generator=gen,
                            worker_init_fn=_seed_worker(args.seed * 1000 + info.rank))
        trainloader = HostLoader(loader, sampler, kin, kgt, gen)
        tf = getattr(db_train, "transform", None) or getattr(getattr(db_train, "dataloaders", [None])[0],
                                                            "transform", None)
        train_tf = [str(t) for t in tf.transforms] if tf is not None else []
    # evaluation visits every validation sample exactly once over all ranks (no wrap-around duplicates)
    val_sampler = DistributedSampler(len(db_val), shuffle=False, drop_last=False, pad=False)
    test_bs = args.testBatch if instance else max(args.testBatch, 1)
    testloader = DataLoader(db_val, batch_size=test_bs, sampler=val_sampler, num_workers=args.workers,
                            pin_memory=pin)
    valloader = None
    if data_path == "gpu" and args.val_path != "cpu":
        # validation on the device too: the same samples (the sampler above), decoded once into a device store
        import copy
        db_val_raw = copy.copy(db_val)
        db_val_raw.transform = None
        valloader = GpuBatchLoader(db_val_raw, val_sampler, test_bs, info.device, args.crop, args.task, seed=args.seed,
                                   rank=info.rank, workers=args.workers, hbm_budget_gb=args.gpu_cache_gb, train=False,
                                   relax=50, zero_pad=bool(args.zero_pad_crop))
    scheduler = LR_Scheduler(args.lr_mode, args.lr, args.nEpochs, len(trainloader))
    # every rank continues ITS OWN augmentation stream (utils.checkpoint writes <ckpt>.state.rank<r> per rank)
    trainloader.load_state_dict(resumed_rank_state.get("loader", {}))

    # parameter report (train_pascal.py:57-70, :156-159, :169)
    p = OrderedDict()
    p["trainBatch"], p["nAveGrad"], p["lr"], p["wd"], p["momentum"] = \
        args.trainBatch, args.nAveGrad, args.lr, args.wd, args.momentum
    p["world_size"], p["model"], p["task"], p["crop"] = info.world_size, args.model, args.task, args.crop
    p["data_path"] = data_path
    p["optimizer"] = str(engine.optimizer)
    p["dataset_train"], p["dataset_test"] = str(db_train), str(db_val)
    p["transformations_train"] = train_tf
    p["transformations_test"] = [str(t) for t in db_val.transform.transforms]
    p["host"], p["started"] = socket.gethostname(), datetime.now().strftime("%b%d_%H-%M-%S")
    if info.is_master:
        helpers.generate_param_report(os.path.join(save_dir, exp_name + ".txt"), p)

    guard = StepGuard(info.device, args.finite_every, args.hang_timeout)
    checker = sanity.BatchSanity(args.task, nclass, 255, normalised=not instance) if sanity.ENABLED else None
    restores = 0
    if info.is_master:
        print("Training Network")
    epoch = args.resume_epoch
    while epoch < args.nEpochs:
        start_time = timeit.default_timer()
        if info.is_master:
            print("Current LR", engine.optimizer.param_groups[0]["lr"])
        try:
            tr_loss, ips, wait_ms = train_epoch(engine, trainloader, scheduler, epoch, args, log, guard, checker)
        except FloatingPointError as e:
            # every rank raised together (StepGuard all-reduces the flag).  "restore": go back to
            # the last checkpoint this run wrote (weights, momentum, RNG) and redo the epoch — at most 3 times.
            if args.on_nonfinite == "restore" and (last_ckpt is None or restores >= 3):
                why = "this run has written no checkpoint yet" if last_ckpt is None else "3 restores already"
                raise FloatingPointError(f"{e} (--on-nonfinite restore: {why})") from e
            if args.on_nonfinite != "restore":
                raise
            restores += 1
            extra = load_checkpoint(last_ckpt, engine.ddp, engine.optimizer)
            engine.ddp.arena.zero_grad()
            best_val = float(extra.get("best_val", best_val))
            # the weights are those of the END of epoch extra["epoch"]: training continues with the epoch after it
            # (LR schedule and sampler permutation included), not with the epoch that failed
            back_to = int(extra.get("epoch", epoch - 1)) + 1
            log.log("restore", epoch=epoch, resumed_epoch=back_to, reason=str(e), checkpoint=last_ckpt,
                    attempt=restores)
            epoch = min(back_to, epoch)
            # do not replay the batches that led to the failure: fresh rank-dependent streams for NumPy's global
            # generator (CPU transforms), the loader's own generator (device path / worker base seeds) and a
            # re-drawn sampler permutation
            np.random.seed(args.seed * 1000 + info.rank + 7919 * restores)
            trainloader.reseed(args.seed, info.rank, restores)
            sampler.seed = args.seed + 104729 * restores
            continue
        log.log("epoch", epoch=epoch, numImages=len(trainloader) * args.trainBatch * info.world_size,
                loss=tr_loss, img_per_s=ips, data_wait_ms=wait_ms, **_memory_report(engine))
        if (epoch % args.snapshot) == args.snapshot - 1 and epoch != 0:
            last_ckpt = os.path.join(save_dir, "models", model_name + "_epoch-" + str(epoch) + ".pth")
            save_checkpoint(last_ckpt, engine.ddp, engine.optimizer, {"epoch": epoch, "best_val": best_val},
                            rank_state={"loader": trainloader.state_dict()})
        guard.progress()
        if args.useTest and epoch % args.nTestInterval == (args.nTestInterval - 1):
            fig = os.path.join(save_dir, "figures", "val_epoch-{}.png".format(epoch))
            t_val = timeit.default_timer()
            if instance:
                if valloader is not None:
                    res = validate_instance_device(engine, valloader, args, epoch, fig, tick=guard.progress)
                else:
                    res = validate_instance(engine, testloader, args, epoch, fig, tick=guard.progress)
                score = max(res["jaccard"])
                log.log("val", epoch=epoch, loss=res["loss"], JACCARD=res["jaccard"],
                        path="device" if valloader is not None else "host",
                        img_per_s=res.get("img_per_s"), seconds=timeit.default_timer() - t_val)
            else:
                if valloader is not None:
                    res = engine.evaluate(_Ticked(valloader, args.max_iters, guard.progress), 21, 255,
                                          graph=_graph_eval(args))
                else:
                    res = engine.evaluate(_KeyLoader(testloader, kin, kgt, args.max_iters, tick=guard.progress), 21, 255,
                                          graph=_graph_eval(args))
                score = res["miou"]
                n_val = (min(args.max_iters, len(testloader)) if args.max_iters else len(testloader)) * test_bs
                log.log("val", epoch=epoch, loss=res["loss"], mIoU=res["miou"], pixel_acc=res["pixel_acc"],
                        path="device" if valloader is not None else "host",
                        img_per_s=n_val * info.world_size / max(timeit.default_timer() - t_val, 1e-9),
                        seconds=timeit.default_timer() - t_val)
                if info.is_master:
                    _semseg_figure(engine, _KeyLoader(testloader, kin, kgt, 1), fig, epoch)
            if score > best_val:
                best_val = score
                last_ckpt = os.path.join(save_dir, "models", model_name + "_best.pth")
                save_checkpoint(last_ckpt, engine.ddp, engine.optimizer, {"epoch": epoch, "best_val": best_val},
                            rank_state={"loader": trainloader.state_dict()})
            log.log("time", epoch=epoch, execution_time=timeit.default_timer() - start_time)
            guard.progress()
        epoch += 1
    guard.close()
    if checker is not None:
        checker.raise_if_flagged()
    save_checkpoint(os.path.join(save_dir, "models", model_name + "_last.pth"), engine.ddp, engine.optimizer,
                    {"epoch": args.nEpochs - 1, "best_val": best_val},
                    rank_state={"loader": trainloader.state_dict()})
    log.close()
    dist_env.shutdown()
    return best_val


if __name__ == "__main__":
    main()
