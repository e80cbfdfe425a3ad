#!/usr/bin/env python
"""Headline benchmark: training images/sec of DeepLabv3+ ResNet-101 at 513×513, bf16, synthetic
VOC-shaped data (21 classes + ignore 255), random-init weights — the metric/config named in
BASELINE.json.  One process per GPU over RCCL/xGMI:

    python bench.py --gpus N --steps K --warmup W        # self-launching: starts the N ranks itself
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W      # same, under torchrun

(the reference drives its GPUs from ONE command too, `/root/reference/train_pascal.py:92`).  With
``--gpus N > 1`` and no ``WORLD_SIZE`` in the environment the parent — before any GPU call — starts
N fresh rank processes (parallel.launch) and exits with their status; rank 0 prints the ONE JSON line.
Multi-rank GPU jobs need ``HSA_ENABLE_IPC_MODE_LEGACY=0`` in the environment before the first HIP
call (dmabuf IPC); the launcher exports it, and ``init_distributed`` refuses to run without it.

Each timed step is the full training step: GPU augmentation (random scale/crop/flip + normalise)
of a device-resident uint8 batch → forward → fused ignore-index CE → backward with overlapped
bucketed gradient all-reduce → fused SGD.  Weak scaling: per-GPU batch is fixed.

``--impl stock`` measures the stock PyTorch-ROCm path (MIOpen conv / ATen BN) on the same model
code — the self-established baseline of BASELINE.md; ``--impl native`` (default) runs the
hand-written HIP kernels.  The stock numbers live in ``bench_baseline.json`` (written by
``--record-baseline``); the ratio reported beside the (null) ``vs_baseline`` is EQUAL-BATCH.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
BASELINE_FILE = os.path.join(ROOT, "bench_baseline.json")   # not under profiles/: that directory is not shipped to GPU runs


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=int(os.environ.get("DPAMD_BENCH_BATCH", "0")),
                    help="per-GPU batch; default 256 for the headline model on the kernels (110 GiB of the 288 GB HBM3E; layers "
                         "whose activations pass 2 GiB run as image chunks), 128 otherwise")
    ap.add_argument("--crop", type=int, default=513)
    ap.add_argument("--model", default="deeplabv3plus_resnet101")
    ap.add_argument("--in-channels", type=int, default=None, help="default: 3 (semseg) / 4 (instance)")
    ap.add_argument("--task", choices=["semseg", "instance"], default="semseg",
                    help="instance: the reference's own workload (`train_pascal.py:86`, `:123-134`) — DANet, 4-channel "
                         "RGB + n-ellipse guidance input, binary object mask, the whole transform chain on the GPU")
    ap.add_argument("--output-stride", type=int, default=None)
    ap.add_argument("--impl", choices=["native", "stock"], default=os.environ.get("DPAMD_BENCH_IMPL", "native"))
    ap.add_argument("--sync-bn", action="store_true")
    ap.add_argument("--bucket-mb", type=int, default=64)
    ap.add_argument("--cpu", action="store_true", help="plumbing run on CPU/gloo (tiny shapes)")
    ap.add_argument("--phases", action="store_true", help="print per-phase HIP-event timings (rank 0)")
    ap.add_argument("--graph-step", action="store_true",
                    help="replay the whole training step from one hipGraph (one rank; pays below ~B32, where the "
                         "host's launches bound the step)")
    ap.add_argument("--record-baseline", action="store_true",
                    help="with --impl stock: store this run in bench_baseline.json")
    ap.add_argument("--source-log", default=None, help="log file name recorded with --record-baseline")
    return ap.parse_args()


def _baseline_key(model: str, crop: int) -> str:
    return f"{model}@{crop}"


def load_self_baseline(model: str, crop: int, batch: int):
    """Equal-batch stock-op record {batch, img_s, source_log, ...} or None (no literals in this file)."""
    try:
        with open(BASELINE_FILE) as f:
            table = json.load(f)
    except (OSError, ValueError):
        return None
    per_batch = table.get(_baseline_key(model, crop), {})
    rec = per_batch.get(str(batch))
    if rec is None and per_batch:
        # no stock run at this batch (its MIOpen search alone passes a GPU call's time limit at 256): the largest
        # measured stock batch, flagged as NOT equal-batch (stock: 252 / 315 / 343 img/s at 32 / 64 / 128)
        nearest = max((int(b) for b in per_batch if int(b) <= batch), default=None)
        if nearest is not None:
            rec = dict(per_batch[str(nearest)], equal_batch=False)
    return rec


def record_self_baseline(model: str, crop: int, batch: int, rec: dict) -> None:
    try:
        with open(BASELINE_FILE) as f:
            table = json.load(f)
    except (OSError, ValueError):
        table = {}
    table.setdefault(_baseline_key(model, crop), {})[str(batch)] = rec
    os.makedirs(os.path.dirname(BASELINE_FILE), exist_ok=True)
    tmp = BASELINE_FILE + ".tmp"
    with open(tmp, "w") as f:
        json.dump(table, f, indent=1, sort_keys=True)
    os.replace(tmp, BASELINE_FILE)


def main():
    args = parse_args()
    # ---- one command, N GPUs: the parent launches the ranks BEFORE anything touches the GPU ----
    from distributedpytorch_amd.parallel.launch import maybe_self_launch
    status = maybe_self_launch(args.gpus, cpu=args.cpu)
    if status is not None:
        sys.exit(status)

    # hang watchdog: DPAMD_HANG_DUMP_S=N dumps every thread's Python stack to stderr
    # and exits if the run is still going after N seconds — a stuck collective names its call site
    import faulthandler
    if os.environ.get("DPAMD_HANG_DUMP_S"):
        faulthandler.dump_traceback_later(int(os.environ["DPAMD_HANG_DUMP_S"]), exit=True)
    else:
        # no knob set: a run that is still going after 15 minutes (a healthy one takes one or two) writes every
        # thread's stack to stderr, every 15 minutes, and keeps running — whoever reads the log of a multi-GPU run that
        # stopped finds out where each rank was
        faulthandler.dump_traceback_later(900, repeat=True, exit=False)
    if os.environ.get("DPAMD_HEARTBEAT_S"):
        # long silent phases (MIOpen's kernel search on the stock path, first-step autotuning): one line a minute
        import threading
        t_start = time.time()

        def beat():
            while True:
                time.sleep(float(os.environ["DPAMD_HEARTBEAT_S"]))
                print(f"[bench] alive, {time.time() - t_start:.0f} s", file=sys.stderr, flush=True)
        threading.Thread(target=beat, daemon=True).start()
    import numpy as np
    import torch
    from distributedpytorch_amd import ops
    from distributedpytorch_amd.models import build_model
    from distributedpytorch_amd.ops import SoftmaxCrossEntropy, affine_augment, make_affine_params
    from distributedpytorch_amd.parallel import TrainEngine, dist_env, init_distributed

    info = init_distributed(device="cpu" if args.cpu else None)
    if info.world_size != args.gpus:
        # never report a job of a different size under the requested label
        raise SystemExit(f"[bench] --gpus {args.gpus} but the job has WORLD_SIZE={info.world_size}")
    on_gpu = info.device.type == "cuda"
    if args.impl == "stock":
        ops.set_backend("torch")
    elif ops.get_backend() != "auto":
        raise SystemExit(f"[bench] --impl native but DPAMD_BACKEND={ops.get_backend()!r} forces the PyTorch ops; "
                         "unset it or pass --impl stock")
    torch.manual_seed(1234)  # identical init on every rank; DDP broadcasts anyway
    instance = args.task == "instance"
    if args.in_channels is None:
        args.in_channels = 4 if instance else 3
    nclass = 1 if instance else 21
    model = build_model(args.model, nclass, output_stride=args.output_stride, sync_bn=args.sync_bn,
                        in_channels=args.in_channels)
    if instance:
        from distributedpytorch_amd.utils import SegmentationMultiLosses
        criterion = SegmentationMultiLosses(1)
    else:
        criterion = SoftmaxCrossEntropy(255)
    engine = TrainEngine(model, criterion, lr=5e-8 if instance else 0.007, momentum=0.9, weight_decay=5e-4,
                         device=info.device, bucket_mb=args.bucket_mb, graph_step=args.graph_step)
    if args.batch <= 0:
        # sized for the 288 GB of HBM3E: 1,037-1,041 img/s at 256 against 1,001 at 128; 320 / 384 / 512 give 1,047 /
        # 1,041-1,059 / 1,034 (profiles/r3_bench_*): flat within the box-to-box spread.  The stock path keeps 128.
        if on_gpu and args.impl == "native" and args.model == "deeplabv3plus_resnet101" and not instance and args.crop == 513:
            args.batch = 256
        elif "xception" in args.model and on_gpu and args.impl == "native":
            args.batch = 128         # output stride 8: 310 img/s at 64, 318 at 96, 317 at 128 (184 GiB) since the
            #                          weight-gradient stream no longer doubles the reserved memory (round 2: 234)
        elif "xception" in args.model:
            args.batch = 64
        elif instance and on_gpu and args.impl == "native":
            args.batch = 64          # DANet 512x512: 366 img/s at 32, 382 at 64 (50 GiB)
        elif instance:
            args.batch = 32
        else:
            args.batch = 128
    B, S = args.batch, args.crop
    g = torch.Generator().manual_seed(100 + info.rank)
    raw_img = torch.randint(0, 256, (B, S, S, 3), dtype=torch.uint8, generator=g).to(info.device)
    # blocky synthetic label map with a band of ignore pixels
    coarse = torch.randint(0, 21, (B, 1, 9, 9), generator=g).float()
    lab = torch.nn.functional.interpolate(coarse, size=(S, S), mode="nearest")[:, 0].to(torch.uint8)
    lab[:, : S // 32] = 255
    raw_lab = lab.to(info.device)
    rng = np.random.default_rng(7 + info.rank)
    extra_ch = None
    if args.in_channels > 3:        # RGB + guidance channels (the instance task's 4-channel input)
        extra_ch = torch.rand(B, args.in_channels - 3, S, S, generator=g).to(info.device)

    if instance:
        # one object per image (an ellipse of random size / position): the binary masks of the instance task
        yy, xx = torch.meshgrid(torch.arange(S), torch.arange(S), indexing="ij")
        cy = torch.randint(S // 4, 3 * S // 4, (B, 1, 1), generator=g)
        cx = torch.randint(S // 4, 3 * S // 4, (B, 1, 1), generator=g)
        ry = torch.randint(S // 10, S // 4, (B, 1, 1), generator=g)
        rx = torch.randint(S // 10, S // 4, (B, 1, 1), generator=g)
        gt_mask = ((((yy - cy).float() / ry) ** 2 + ((xx - cx).float() / rx) ** 2) <= 1.0).to(torch.uint8)
        gt_mask = gt_mask.contiguous().to(info.device)
        slot_idx = torch.arange(B, dtype=torch.int32, device=info.device)
        sizes_host = np.full((B, 2), S, dtype=np.int32)
        sizes_dev = torch.from_numpy(sizes_host).to(info.device)
        from distributedpytorch_amd.ops import instance_augment, make_warp_params

    def one_step():
        if instance:
            # the reference's transform chain (flip, scale+rotate, mask crop, resize, guidance map, concat) on device
            warp = make_warp_params(B, sizes_host, (-20, 20), (.75, 1.25), 0.5, rng)
            with engine.timer.phase("augment"):
                x, y, _ = instance_augment(raw_img, gt_mask, slot_idx, sizes_dev,
                                           torch.from_numpy(warp).to(info.device, non_blocking=True), (S, S), 50, True,
                                           seed=int(rng.integers(0, 2 ** 31 - 1)), out_dtype=engine.compute_dtype,
                                           validated=True)
            return engine.train_step(x, y)
        mats = make_affine_params(B, (S, S), (S, S), scales=(0.5, 2.0), flip_p=0.5, rng=rng)
        with engine.timer.phase("augment"):
            x, y = affine_augment(raw_img, raw_lab, torch.from_numpy(mats).to(info.device, non_blocking=True),
                                  (S, S), out_dtype=engine.compute_dtype)
            if extra_ch is not None:
                x = torch.cat([x, extra_ch.to(x.dtype)], dim=1).contiguous(memory_format=torch.channels_last)
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
    phases = engine.timer.summary()
    # one more step OUTSIDE the timed region with the per-bucket timeline on (parallel.ddp.profile_next_step): when
    # each bucket could start, when it was done, when backward ended — which collective is exposed, and by how much
    comm_prof = None
    if on_gpu and engine.ddp._comm:
        engine.ddp.profile_next_step()
        one_step()
        sync()
        comm_prof = engine.ddp.comm_profile()
        if comm_prof is not None:
            comm_prof["exposed_after_backward_ms_max_over_ranks"] = round(dist_env.all_reduce_scalar(
                comm_prof["exposed_after_backward_ms"], "max", info.device), 3)
    # exposed communication = what the step's stream waited for in finish_gradient_sync (HIP events), worst rank
    comm_wait = dist_env.all_reduce_scalar(phases.get("comm_wait", 0.0), "max", info.device if on_gpu else None)
    devices = [info.device.index if on_gpu else -1]
    if info.world_size > 1:
        import torch.distributed as dist
        gathered = [None] * info.world_size
        dist.all_gather_object(gathered, devices[0])
        devices = gathered
    world = info.world_size
    ips = world * B * args.steps / elapsed
    if info.is_master:
        if args.phases:
            print("[bench] phases(ms):", {k: round(v, 3) for k, v in phases.items()}, file=sys.stderr)
            if on_gpu:
                print(f"[bench] peak HBM {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB", file=sys.stderr)
        headline = (args.model == "deeplabv3plus_resnet101" and args.crop == 513 and args.in_channels == 3
                    and not instance)
        rec = {
            "metric": ("train images/sec (node) DeepLabv3+ R101 513x513" if headline
                       else f"train images/sec (node) {args.model} {args.crop}x{args.crop}"),
            "value": round(ips, 3), "unit": "images/sec", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(1e3 * elapsed / args.steps, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "bf16" if on_gpu else "fp32", "data": "synthetic",
            "config": {"model": args.model, "global_batch": world * B, "per_gpu_batch": B,
                       "seq_len": S, "crop": S, "num_classes": nclass, "in_channels": args.in_channels,
                       "task": args.task, "parallelism": f"dp{world}", "impl": args.impl, "sync_bn": bool(args.sync_bn),
                       "optimizer": "sgd-momentum", "final_loss": round(final_loss, 4)},
            "comm": {"backend": info.backend, "rccl_world": world if info.backend == "nccl" else 0,
                     "devices": devices, "buckets": len(engine.ddp.arena.buckets), "bucket_mb": args.bucket_mb,
                     "ddp_mode": engine.ddp.mode, "comm_wait_ms": round(comm_wait, 3),
                     "exposed_comm_ms": round(comm_wait, 3)},
            "runtime": {"ops_backend": ops.get_backend(), "graph_step": bool(engine.graph_step),
                        "graphs_captured": len(engine._graphs)},
        }
        if on_gpu:
            rec["runtime"]["hbm_peak_gib"] = round(torch.cuda.max_memory_allocated() / 2 ** 30, 1)
            rec["runtime"]["hbm_reserved_peak_gib"] = round(torch.cuda.max_memory_reserved() / 2 ** 30, 1)
            # > 0: the caching allocator ran out, freed its cache (a device synchronisation) and retried — the step
            # time then measures the allocator, not the kernels: lower the batch or set expandable segments
            rec["runtime"]["alloc_retries"] = int(torch.cuda.memory_stats().get("num_alloc_retries", 0))
        if args.impl == "stock" and os.environ.get("MIOPEN_FIND_MODE"):
            rec["runtime"]["miopen_find_mode"] = os.environ["MIOPEN_FIND_MODE"]
        if comm_prof is not None:
            rec["comm"]["timeline_rank0"] = comm_prof
        mid = getattr(getattr(model, "backbone", None), "mid_physical", None)
        if mid is not None:     # Xception-65: the 728-channel flow stored 768 wide (zero padding channels, same network)
            rec["config"]["mid_channels"] = {"logical": 728, "stored": mid}
        if on_gpu and args.impl == "native":
            from distributedpytorch_amd.ops import _backend, _hip_conv
            bi = _backend.build_info()
            tr = _hip_conv.tune_report()
            rec["runtime"].update(lib_hash=bi.get("lib_hash"), lib_matches_sources=bi.get("lib_hash") == bi.get("source_hash"),
                                  lib_rebuilt_at_load=bi.get("rebuilt"), tile_table_entries=tr["packaged"],
                                  tiles_autotuned_in_run=tr["measured_here"],
                                  tile_table_hit=tr["measured_here"] == 0)
            from distributedpytorch_amd.ops import bnfold
            rec["runtime"]["bn3_fold_backwards"] = dict(bnfold.COUNTS)
            rec["runtime"]["deterministic"] = _backend.deterministic()
        if on_gpu:
            base = load_self_baseline(args.model, S, B)
            if args.impl == "native" and base is not None:
                # the reference publishes nothing (vs_baseline stays null); the self-established
                # baseline is the stock PyTorch-ROCm op path, same model code, same GPU type, SAME per-GPU batch
                ratio = round(ips / world / base["img_s"], 3)
                if base.get("equal_batch", True):
                    rec["self_baseline"] = dict(base, ratio_per_gpu_equal_batch=ratio)
                else:
                    rec["self_baseline"] = dict(base, **{f"ratio_per_gpu_vs_stock_at_batch_{base['batch']}": ratio})
            if args.impl == "stock" and args.record_baseline and world == 1:
                record_self_baseline(args.model, S, B, {
                    "batch": B, "img_s": round(ips, 2), "ms_per_step": round(1e3 * elapsed / args.steps, 2),
                    "steps": args.steps, "warmup": args.warmup, "source_log": args.source_log,
                    "find_mode": os.environ.get("MIOPEN_FIND_MODE", "default (exhaustive search)"),
                    "note": "stock PyTorch-ROCm ops (MIOpen conv, ATen BN/CE, full-resolution logits + ATen "
                            "bilinear upsample), same model code, 1 x MI355X"})
        print(json.dumps(rec), flush=True)
    dist_env.shutdown()


if __name__ == "__main__":
    main()
