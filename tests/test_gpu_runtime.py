"""Round-2 GPU tests: device data-path kernels (csrc/augment.hip) against their plain-torch references, the
RCCL choreography on a 1-rank group, and — when the box has >= 2 GPUs — the real multi-GPU path over the
**nccl** (= RCCL) backend: DDP gradients == single-process large batch, bit-identical replicas under
DPAMD_DEBUG=1, SyncBN over RCCL and over the one-shot p2p kernel across DIFFERENT devices, reduce-scatter +
all-gather bucket mode.  The multi-GPU tests skip cleanly on a 1-GPU box."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pytestmark = pytest.mark.gpu


def _ngpu() -> int:
    return torch.cuda.device_count()


def _free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _scene(h, w, seed):
    from distributedpytorch_amd.data.pascal import _synthetic_scene
    return _synthetic_scene(np.random.default_rng(seed), h, w)


# ------------------------------------------------------------------------------ data-path kernels
@pytest.mark.parametrize("interp", ["bilinear", "bicubic"])
def test_affine_augment2_kernel_matches_reference(interp):
    from distributedpytorch_amd import ops
    n, hs, ws = 5, 72, 88
    g = torch.Generator().manual_seed(0)
    store = torch.randint(0, 256, (n, hs, ws, 3), dtype=torch.uint8, generator=g)
    lab = torch.randint(0, 21, (n, hs, ws), dtype=torch.uint8, generator=g)
    sizes_all = torch.tensor([[72, 88], [50, 61], [72, 40], [33, 88], [64, 64]], dtype=torch.int32)
    src = torch.tensor([3, 0, 4, 1], dtype=torch.int32)
    sizes = sizes_all[src.long()]
    rng = np.random.default_rng(1)
    mats = torch.from_numpy(ops.make_affine_params(4, sizes.numpy(), (65, 57), (0.5, 2.0), (-20, 20), 0.5, rng))
    clip = torch.tensor([[1.0, 0, -3, 0, 1, -2, 50, 40]] * 4)
    for use_clip in (None, clip):
        ref_x, ref_y = ops.affine_augment2(store, lab, mats, (65, 57), src, sizes, use_clip, interp=interp,
                                           clamp=interp == "bicubic", out_dtype=torch.float32)
        x, y = ops.affine_augment2(store.cuda(), lab.cuda(), mats.cuda(), (65, 57), src.cuda(), sizes.cuda(),
                                   None if use_clip is None else use_clip.cuda(), interp=interp,
                                   clamp=interp == "bicubic")
        assert x.is_contiguous(memory_format=torch.channels_last) and x.dtype == torch.bfloat16
        err = (x.float().cpu() - ref_x).abs()
        assert err.max() < 0.06 and err.mean() < 6e-3, (err.max(), err.mean())      # bf16 output of O(2) values
        assert (y.cpu() != ref_y).float().mean() < 2e-3                              # nearest ties


def test_instance_pipeline_kernels_match_reference():
    """bbox / crop composition / extreme points / guidance map: each HIP kernel against its torch reference, then
    the whole chain (`/root/reference/train_pascal.py:123-134`)."""
    from distributedpytorch_amd import ops
    hs, ws = 96, 120
    imgs, gts = [], []
    for s in range(4):
        img, _, inst = _scene(hs, ws, 10 + s)
        imgs.append(torch.from_numpy(img))
        gts.append(torch.from_numpy((inst == 1).astype(np.uint8)))
    gts[3].zero_()                                                  # an empty object: all-zero crop, no NaN
    store, gstore = torch.stack(imgs), torch.stack(gts)
    src = torch.tensor([2, 0, 3, 1], dtype=torch.int32)
    sizes = torch.tensor([[96, 120], [96, 120], [96, 120], [80, 100]], dtype=torch.int32)
    warp = torch.from_numpy(ops.make_warp_params(4, sizes.numpy(), (-20, 20), (.75, 1.25), 0.5,
                                                 np.random.default_rng(2)))
    bb_ref = ops.mask_bbox(gstore, warp, src, sizes)
    bb = ops.mask_bbox(gstore.cuda(), warp.cuda(), src.cuda(), sizes.cuda())
    assert torch.equal(bb.cpu()[bb_ref[:, 2] >= 0], bb_ref[bb_ref[:, 2] >= 0])
    assert int(bb.cpu()[2, 2]) == -1
    m_ref, c_ref, b_ref = ops.compose_crop(warp, bb_ref, (64, 64), 10, True, sizes, (hs, ws))
    m, c, b = ops.compose_crop(warp.cuda(), bb, (64, 64), 10, True, sizes.cuda(), (hs, ws))
    assert torch.allclose(m.cpu(), m_ref, rtol=1e-5, atol=1e-4) and torch.allclose(c.cpu(), c_ref, atol=1e-4)
    assert torch.equal(b.cpu(), b_ref)
    xr, yr, _ = ops.instance_augment(store, gstore, src, sizes, warp, (64, 64), 10, True, random_points=False,
                                     out_dtype=torch.float32)
    x, y, _ = ops.instance_augment(store.cuda(), gstore.cuda(), src.cuda(), sizes.cuda(), warp.cuda(), (64, 64), 10,
                                   True, random_points=False)
    assert x.shape == (4, 4, 64, 64) and x.is_contiguous(memory_format=torch.channels_last)
    assert (y.cpu() != yr).float().mean() < 3e-3
    d = (x.float().cpu() - xr).abs()
    assert d[:, :3].mean() < 0.6 and d[:, :3].max() < 6.0          # bf16 storage of 0..255 values: 1 ulp = 1.0 at 255
    pts_ref = ops.extreme_points((yr[:, 0] > 0).to(torch.uint8), random=False)
    pts = ops.extreme_points((y[:, 0] > 0).to(torch.uint8).contiguous(), random=False)
    same_mask = [(y[i].cpu() == yr[i]).all().item() for i in range(4)]
    for i in range(4):
        if same_mask[i]:
            assert torch.equal(pts[i].cpu(), pts_ref[i])
    assert (pts[[i for i in range(4) if int(src[i]) == 3]] == -1).all()
    # guidance kernel alone, same points on both sides
    out = torch.zeros(4, 4, 64, 64, device="cuda", dtype=torch.bfloat16).contiguous(memory_format=torch.channels_last)
    ops.nellipse_guidance_(out, pts, 3, 0.6)
    ref = torch.zeros(4, 4, 64, 64)
    ops.nellipse_guidance_(ref, pts.cpu(), 3, 0.6)
    assert (out[:, 3].float().cpu() - ref[:, 3]).abs().max() < 1.5   # bf16 of 0..255
    assert float(out[:, :3].abs().max()) == 0.0                      # other channels untouched
    # random extreme points: always ON the object and on its extreme lines
    rp = ops.extreme_points((y[:, 0] > 0).to(torch.uint8).contiguous(), random=True, seed=5).cpu()
    for i in range(4):
        mk = (y[i, 0] > 0).cpu().numpy()
        if mk.any():
            ys_, xs_ = np.where(mk)
            p = rp[i].numpy()
            assert p[0, 0] == xs_.min() and p[1, 0] == xs_.max() and p[2, 1] == ys_.min() and p[3, 1] == ys_.max()
            assert all(mk[q[1], q[0]] for q in p)


def test_paste_jaccard_kernel_matches_the_host_formulation():
    """csrc/augment.hip paste_jaccard_kernel against ops.paste_jaccard on CPU tensors (helpers.crop2fullmask +
    calc_jaccard of the reference's validation loop, `/root/reference/train_pascal.py:280-306`): smooth random
    logits, boxes that reach outside the image, void pixels, an empty box, slots picked through src_idx, fp32 / bf16."""
    from distributedpytorch_amd.ops.augment import paste_jaccard
    g = torch.Generator().manual_seed(2)
    nslots, hs, ws, bsz, hc = 6, 90, 120, 5, 64
    gt = (torch.rand(nslots, hs, ws, generator=g) > 0.6).to(torch.uint8)
    void = (torch.rand(nslots, hs, ws, generator=g) > 0.9).to(torch.uint8)
    # smooth logits: a coarse random field upsampled, so that thresholds cut contours and not salt and pepper
    logits = torch.nn.functional.interpolate(torch.randn(bsz, 1, 8, 8, generator=g) * 3, size=(hc, hc), mode="bicubic")
    src = torch.tensor([4, 0, 5, 2, 2], dtype=torch.int32)
    sizes = torch.tensor([[90, 120], [70, 100], [90, 64], [33, 120], [90, 120]], dtype=torch.int32)
    bbox = torch.tensor([[-20, -30, 100, 80], [10, 5, 80, 60], [0, 0, 63, 89], [0, 0, -1, -1], [30, 20, 150, 110]],
                        dtype=torch.int32)
    for relax in (0, 7):
        ref = paste_jaccard(logits, bbox, gt, void, src, sizes, relax, (0.3, 0.5, 0.8))
        for dt in (torch.float32, torch.bfloat16):
            lg = logits.to(dt)
            want = ref if dt == torch.float32 else paste_jaccard(lg.float(), bbox, gt, void, src, sizes, relax,
                                                                 (0.3, 0.5, 0.8))
            got = paste_jaccard(lg.cuda(), bbox.cuda(), gt.cuda(), void.cuda(), src.cuda(), sizes.cuda(), relax,
                                (0.3, 0.5, 0.8)).cpu()
            assert got.shape == want.shape == (bsz, 3, 2)
            assert got[3].abs().sum() == 0 or int(want[3].abs().sum()) == int(got[3].abs().sum())
            # a pixel whose interpolated probability sits within float rounding of a threshold may flip: a handful
            diff = (got - want).abs()
            assert int(diff.max()) <= 6 and float(diff.float().mean()) < 1.0, (relax, dt, got.tolist(), want.tolist())
            assert int(want[:, :, 1].sum()) > 10000
    no_void = paste_jaccard(logits.cuda(), bbox.cuda(), gt.cuda(), None, src.cuda(), sizes.cuda(), 0, (0.5,)).cpu()
    assert (no_void[:, 0, 1] >= ref[:, 1, 1] - 6).all()


@pytest.mark.parametrize("task", ["instance", "semseg"])
def test_entrypoint_validates_on_the_device(task, tmp_path):
    """with --data-path gpu the entrypoint's validation runs on the device too (device sample store,
    device transform chain, paste_jaccard / confusion-matrix kernels, one host read per epoch) and logs its
    throughput; the numbers equal the host validation (--val-path cpu) of the same weights."""
    import json

    import train_pascal as tp
    from distributedpytorch_amd import ops
    model = "danet_resnet50" if task == "instance" else "deeplabv3plus_resnet18"
    common = ["--task", task, "--synthetic", "--model", model, "--crop", "128" if task == "instance" else "129",
              "--nEpochs", "1", "--trainBatch", "4", "--testBatch", "4", "--synthetic-len", "48", "--workers", "0",
              "--log-every", "0", "--max-iters", "3", "--seed", "3"]
    vals = {}
    with ops.deterministic_mode(True):            # the same weights after the same three steps on both runs
        for path in ("auto", "cpu"):
            d = tmp_path / path
            tp.main(common + ["--save-dir", str(d), "--val-path", path])
            recs = [json.loads(l) for l in (d / "run_0" / "log.jsonl").read_text().splitlines()]
            vals[path] = next(r for r in recs if r.get("kind") == "val")
    assert vals["auto"]["path"] == "device" and vals["cpu"]["path"] == "host"
    assert vals["auto"]["img_per_s"] > 0
    if task == "instance":
        for a, b in zip(vals["auto"]["JACCARD"], vals["cpu"]["JACCARD"]):
            assert abs(a - b) < 5e-3, vals
    else:
        assert abs(vals["auto"]["mIoU"] - vals["cpu"]["mIoU"]) < 5e-3, vals


def test_device_forms_of_the_unused_transforms_on_the_gpu():
    """ops.bbox_mask / extreme_points_heatmap / confidence_map / dynamic_crop_relax on device masks (bounding boxes and
    extreme points from the HIP kernels) equal the same functions on CPU tensors, which tests/test_runtime.py checks
    against data/transforms.py (CreateBBMask, ExtremePoints, AddConfidenceMap, CropFromMask)."""
    from distributedpytorch_amd import ops
    g = torch.Generator().manual_seed(7)
    masks = torch.zeros(6, 90, 120, dtype=torch.uint8)
    yy, xx = torch.meshgrid(torch.arange(90), torch.arange(120), indexing="ij")
    for b in range(5):
        cy, cx = int(torch.randint(20, 70, (1,), generator=g)), int(torch.randint(30, 90, (1,), generator=g))
        ry, rx = int(torch.randint(6, 20, (1,), generator=g)), int(torch.randint(8, 30, (1,), generator=g))
        masks[b] = (((yy - cy) / ry) ** 2 + ((xx - cx) / rx) ** 2 <= 1).to(torch.uint8)
    dm = masks.cuda()
    assert torch.equal(ops.bbox_mask(dm).cpu(), ops.bbox_mask(masks))
    assert torch.allclose(ops.extreme_points_heatmap(dm).cpu(), ops.extreme_points_heatmap(masks), atol=1e-5)
    for kind in ("l1l2", "gaussian"):
        a, b = ops.confidence_map(dm, kind).cpu(), ops.confidence_map(masks, kind)
        assert torch.allclose(a, b, atol=0.05), (kind, float((a - b).abs().max()))
    bd, bc = ops.augment._identity_bbox(dm), ops.augment._identity_bbox(masks)
    assert torch.equal(bd.cpu(), bc)
    assert torch.equal(ops.dynamic_crop_relax(bd).cpu(), ops.dynamic_crop_relax(bc))
    assert torch.equal(ops.expand_bbox(bd, ops.dynamic_crop_relax(bd)).cpu(), ops.expand_bbox(bc, ops.dynamic_crop_relax(bc)))


def test_checkpoint_host_copies_are_batched_and_exact():
    """utils.hostcopy: the checkpoint's tensors leave the device in one copy per dtype (1,021 blocking copies per R101
    checkpoint before) — values, shapes and contiguity as `.cpu().contiguous()` gives them, storages independent;
    Xception's padded flow still saves in its logical 728-wide shapes."""
    from distributedpytorch_amd.models import build_model
    from distributedpytorch_amd.ops import SoftmaxCrossEntropy
    from distributedpytorch_amd.parallel import TrainEngine
    from distributedpytorch_amd.utils.checkpoint import reference_state_dict
    dev = torch.device("cuda", 0)
    torch.manual_seed(3)
    model = build_model("deeplabv3plus_xception", 21, output_stride=16)
    eng = TrainEngine(model, SoftmaxCrossEntropy(255), lr=0.01, momentum=0.9, device=dev)
    x = torch.randn(2, 3, 97, 97, device=dev).to(eng.compute_dtype).contiguous(memory_format=torch.channels_last)
    y = torch.randint(0, 21, (2, 97, 97), device=dev, dtype=torch.uint8)
    eng.train_step(x, y)
    sd = reference_state_dict(eng.ddp)
    ref = {"module." + k: (v.detach().float() if v.is_floating_point() else v.detach()).cpu().contiguous()
           for k, v in eng.ddp.module.state_dict().items()}
    assert list(sd) == list(ref)
    ptrs = set()
    for k in ref:
        assert sd[k].device.type == "cpu" and sd[k].is_contiguous() and sd[k].shape == ref[k].shape, k
        assert sd[k].dtype == ref[k].dtype and torch.equal(sd[k], ref[k]), k
        ptrs.add(sd[k].untyped_storage().data_ptr())
    assert len(ptrs) == sum(1 for v in ref.values() if v.numel())
    assert max(v.shape[0] for k, v in sd.items() if ".middle." in k and v.dim() == 4) == 728       # logical widths
    mom = eng.optimizer.state_dict()["momentum_by_name"]
    assert any(float(v.abs().max()) > 0 for v in mom.values())
    for s in eng.optimizer.arena.slots:
        assert mom[s.name].device.type == "cpu" and mom[s.name].is_contiguous()
    eng2 = TrainEngine(build_model("deeplabv3plus_xception", 21, output_stride=16), SoftmaxCrossEntropy(255), lr=0.01,
                       momentum=0.9, device=dev)
    eng2.optimizer.load_state_dict(eng.optimizer.state_dict())
    mom2 = eng2.optimizer.state_dict()["momentum_by_name"]
    assert list(mom2) == list(mom) and all(torch.equal(mom2[k], mom[k]) for k in mom)
    differ = int((eng2.optimizer.arena.flat_m != eng.optimizer.arena.flat_m).sum())
    print("arena positions that differ after the round trip (padding channels only):", differ)


def test_gpu_batch_loader_feeds_engine():
    """train_pascal's GPU data path: device sample store -> gather -> TrainEngine (both tasks)."""
    from distributedpytorch_amd.data import SyntheticVOC
    from distributedpytorch_amd.data.gpu_pipeline import GpuBatchLoader
    from distributedpytorch_amd.models import build_model
    from distributedpytorch_amd.ops import SoftmaxCrossEntropy
    from distributedpytorch_amd.parallel import DistributedSampler, TrainEngine
    dev = torch.device("cuda", 0)
    db = SyntheticVOC(8, (97, 97), None, "semantic", seed=1)
    ld = GpuBatchLoader(db, DistributedSampler(len(db), 1, 0, shuffle=True), 4, dev, 97, "semseg", workers=0)
    ld.set_epoch(0)
    torch.manual_seed(0)
    eng = TrainEngine(build_model("deeplabv3plus_resnet18", 21), SoftmaxCrossEntropy(255), lr=0.01, device=dev)
    losses = [float(eng.train_step(x, y)) for x, y in ld]
    assert len(losses) == 2 and all(np.isfinite(losses))
    assert ld.store is not None and ld.store.images.is_cuda and ld.store.filled.all()
    ld2 = GpuBatchLoader(db, DistributedSampler(len(db), 1, 0, shuffle=True), 4, dev, 97, "semseg", workers=0,
                         mode="stream")
    xs = [x for x, _ in ld2]
    assert len(xs) == 2 and xs[0].is_cuda and xs[0].dtype == torch.bfloat16


# ------------------------------------------------------------------------------ RCCL choreography, one rank
def _force_worker(rank, port, mode):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0",
                      HSA_ENABLE_IPC_MODE_LEGACY="0", DPAMD_DDP_FORCE="1", DPAMD_DDP_MODE=mode)
    from distributedpytorch_amd.models import build_model
    from distributedpytorch_amd.ops import SoftmaxCrossEntropy
    from distributedpytorch_amd.parallel import TrainEngine, debug, dist_env, init_distributed
    debug.ENABLED = True
    info = init_distributed()
    try:
        assert info.backend == "nccl" and dist.is_initialized()
        torch.manual_seed(3)
        eng = TrainEngine(build_model("deeplabv3plus_resnet18", 21, sync_bn=True), SoftmaxCrossEntropy(255), lr=0.01,
                          device=info.device, bucket_mb=4)
        assert eng.ddp._comm and len(eng.ddp.arena.buckets) >= 3
        g = torch.Generator().manual_seed(5)
        x = torch.randn(2, 3, 97, 97, generator=g).to(info.device)
        y = torch.randint(0, 21, (2, 97, 97), generator=g).to(info.device)
        losses = [float(eng.train_step(x, y)) for _ in range(4)]
        assert all(np.isfinite(losses)) and losses[-1] < losses[0]
        assert eng.ddp.stats["buckets_launched"] == 4 * len(eng.ddp.arena.buckets)
        assert eng.ddp._comm_stream is not None            # collectives were issued from the comm-launch stream
    finally:
        dist_env.shutdown()


@pytest.mark.parametrize("mode", ["allreduce", "rs_ag"])
def test_rccl_choreography_on_one_rank(mode):
    """Flat broadcast, hook-launched async bucket collectives from the comm-launch stream (events on the step's and
    the weight-gradient stream), SyncBN collectives, high-priority RCCL stream — all on a real 1-rank RCCL group."""
    mp.spawn(_force_worker, args=(_free_port(), mode), nprocs=1, join=True)


# ------------------------------------------------------------------------------ real multi-GPU (nccl)
def _nccl_worker(rank, world, port, case):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank), HSA_ENABLE_IPC_MODE_LEGACY="0")
    if case == "rs_ag":
        os.environ["DPAMD_DDP_MODE"] = "rs_ag"
    from distributedpytorch_amd import ops
    from distributedpytorch_amd.models import build_model
    from distributedpytorch_amd.ops import SoftmaxCrossEntropy
    from distributedpytorch_amd.parallel import TrainEngine, debug, dist_env, init_distributed, p2p
    debug.ENABLED = True
    p2p.ENABLED = case == "syncbn_p2p"
    p2p.CHECK_EVERY = 1
    info = init_distributed()
    try:
        assert info.backend == "nccl" and info.device.index == rank           # one DIFFERENT device per rank
        dev = info.device
        per = 2
        g = torch.Generator().manual_seed(50)
        xs = torch.randn(per * world, 3, 97, 97, generator=g)
        ys = torch.randint(0, 21, (per * world, 97, 97), generator=g)
        ys[:, :3] = 255
        if case in ("grads", "rs_ag"):
            # DDP gradients == single-process large-batch gradients (frozen BN: local statistics would differ)
            torch.manual_seed(7 + rank)                                       # the broadcast must fix this
            eng = TrainEngine(build_model("deeplabv3plus_resnet18", 21, freeze_bn=True), SoftmaxCrossEntropy(255),
                              lr=0.0, momentum=0.0, weight_decay=0.0, device=dev, bucket_mb=4)
            arena = eng.ddp.arena
            eng.model.train()
            x = xs[per * rank:per * (rank + 1)].to(dev)
            y = ys[per * rank:per * (rank + 1)].to(dev)
            out = eng.ddp(eng.prepare_input(x))
            eng.criterion(out, y).backward()
            scale = eng.ddp.finish_gradient_sync()
            got = (arena.flat_g * scale).clone()
            torch.cuda.synchronize()
            # the same weights (rank 0's, after the broadcast), whole batch, one process
            ref_model = build_model("deeplabv3plus_resnet18", 21, freeze_bn=True)
            ref_eng = TrainEngine(ref_model, SoftmaxCrossEntropy(255), lr=0.0, momentum=0.0, weight_decay=0.0, device=dev,
                                  bucket_mb=4, process_group=None)
            ref_eng.ddp._comm = False
            ref_eng.ddp.arena.flat_p.copy_(arena.flat_p)
            ref_eng.ddp.arena.refresh_lowp()
            ref_eng.model.train()
            out = ref_eng.ddp(ref_eng.prepare_input(xs.to(dev)))
            ref_eng.criterion(out, ys.to(dev)).backward()
            ref_eng.ddp.finish_gradient_sync()
            ref = ref_eng.ddp.arena.flat_g
            cos = torch.nn.functional.cosine_similarity(got, ref, dim=0)
            rel = (got - ref).norm() / ref.norm()
            assert cos > 0.999 and rel < 0.03, (float(cos), float(rel))      # bf16 kernels, different batch split
        else:
            sync_bn = case in ("syncbn_rccl", "syncbn_p2p")
            torch.manual_seed(7 + rank)
            eng = TrainEngine(build_model("deeplabv3plus_resnet18", 21, sync_bn=sync_bn), SoftmaxCrossEntropy(255),
                              lr=0.01, device=dev, bucket_mb=4)
            x = xs[per * rank:per * (rank + 1)].to(dev)
            y = ys[per * rank:per * (rank + 1)].to(dev)
            losses = [float(eng.train_step(x, y)) for _ in range(4)]
            assert all(np.isfinite(losses))
            flat = eng.ddp.arena.flat_p.detach().clone()
            other = flat.clone()
            dist.broadcast(other, src=0)
            assert torch.equal(flat, other), "replicas diverged"            # bit-identical after 4 steps
            if sync_bn:
                # SyncBN == BN over the concatenated batch (a single layer, full-batch reference on this rank)
                torch.manual_seed(3)
                full = torch.randn(per * world, 64, 17, 17, device=dev).to(torch.bfloat16)
                bn = ops.BatchNorm2d(64, sync=True).to(dev).train()
                bn_ref = ops.BatchNorm2d(64).to(dev).train()
                xi = full[per * rank:per * (rank + 1)].contiguous(memory_format=torch.channels_last).requires_grad_(True)
                xr = full.contiguous(memory_format=torch.channels_last).requires_grad_(True)
                yi, yr = bn(xi, relu=True), bn_ref(xr, relu=True)
                assert torch.allclose(yi.float(), yr[per * rank:per * (rank + 1)].float(), atol=2e-2, rtol=2e-2)
                yi.float().pow(2).sum().backward()
                yr.float().pow(2).sum().backward()
                assert torch.allclose(xi.grad.float(), xr.grad[per * rank:per * (rank + 1)].float(), atol=3e-2, rtol=3e-2)
                assert torch.allclose(bn.running_mean, bn_ref.running_mean, atol=1e-3)
                if case == "syncbn_p2p":
                    assert p2p._default is not None, "the one-shot p2p kernel was not used"
                    p2p._default.check()
    finally:
        if p2p._default is not None:
            p2p._default.close()
            p2p._default = None
        dist_env.shutdown()


@pytest.mark.skipif(_ngpu() < 2, reason="needs >= 2 GPUs (RCCL refuses two ranks on one device)")
@pytest.mark.parametrize("case", ["grads", "replicas", "syncbn_rccl", "syncbn_p2p", "rs_ag"])
def test_multi_gpu_nccl(case):
    world = min(_ngpu(), 4)
    mp.spawn(_nccl_worker, args=(world, _free_port(), case), nprocs=world, join=True)


@pytest.mark.skipif(_ngpu() < 2, reason="needs >= 2 GPUs")
def test_bench_self_launch_multi_gpu():
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--model",
                        "deeplabv3plus_resnet18", "--crop", "129", "--batch", "4", "--steps", "3", "--warmup", "2"],
                       capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    rec = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert rec["n_gpus"] == 2 and rec["comm"]["rccl_world"] == 2 and sorted(rec["comm"]["devices"]) == [0, 1]
