"""CPU tests for the model zoo, the training engine, checkpoint compatibility and the small
utilities.  The reference ships no models (they live in external
packages, `/root/reference/train_pascal.py:31-33`), so the model checks pin *contracts* it relies
on: DANet returns a 3-tuple (`:258-260`), outputs are input-sized, ``get_1x/10x_lr_params`` exist
(`:90-91`), and ``state_dict`` keys carry a ``module.`` prefix after wrapping (`:92`, `:103`).
"""
import json
import math
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from distributedpytorch_amd import ops  # noqa: E402
from distributedpytorch_amd.models import MODEL_NAMES, build_model  # noqa: E402
from distributedpytorch_amd.parallel import (DistributedDataParallel, FlatSGD, ParamArena,  # noqa: E402
                                             TrainEngine)
from distributedpytorch_amd.utils import (JsonlLogger, LR_Scheduler, SegmentationMultiLosses,  # noqa: E402
                                          class_balanced_cross_entropy_loss, load_checkpoint,
                                          load_weights, reference_state_dict, save_checkpoint)


# --------------------------------------------------------------------------------------
# models
# --------------------------------------------------------------------------------------
@pytest.mark.parametrize("name,size", [
    ("fcn_resnet18", 64), ("deeplabv3plus_resnet18", 65), ("deeplabv3plus_resnet50", 65),
    ("deeplabv3plus_xception65", 65), ("danet_resnet50", 64),
])
def test_model_forward_backward_shapes(name, size):
    torch.manual_seed(0)
    nclass = 5
    m = build_model(name, nclass)
    x = torch.randn(2, 3, size, size)
    out = m(x)
    outs = out if isinstance(out, (tuple, list)) else (out,)
    if name.startswith("danet"):
        assert len(outs) == 3          # (fused, position-attention, channel-attention) heads
    for o in outs:
        assert o.shape == (2, nclass, size, size)
    loss = sum(o.float().mean() for o in outs)
    loss.backward()
    got = [p.grad is not None for p in m.parameters()]
    assert sum(got) >= 0.9 * len(got)
    one = {id(p) for p in m.get_1x_lr_params()}
    ten = {id(p) for p in m.get_10x_lr_params()}
    assert one and ten and not (one & ten)
    assert one | ten == {id(p) for p in m.parameters()}


def test_model_registry_and_aliases():
    assert "deeplabv3plus_resnet101" in MODEL_NAMES and "deeplabv3plus_xception65" in MODEL_NAMES
    with pytest.raises(ValueError):
        build_model("nonsense_net")
    m = build_model("fcn_r18", 3, sync_bn=True, freeze_bn=True)
    bns = [b for b in m.modules() if isinstance(b, ops.BatchNorm2d)]
    assert bns and all(b.sync and b.frozen for b in bns)
    assert all(not b.weight.requires_grad for b in bns)


def test_deeplab_r101_param_count_and_keys():
    """~59 M parameters and torchvision-style backbone key names."""
    m = build_model("deeplabv3plus_resnet101", 21)
    n = sum(p.numel() for p in m.parameters())
    assert 55e6 < n < 65e6, n
    keys = set(m.state_dict().keys())
    for k in ("backbone.conv1.weight", "backbone.layer1.0.downsample.0.weight", "backbone.layer3.22.bn3.running_var",
              "aspp.aspp2.atrous_conv.weight", "aspp.global_avg_pool.1.weight", "decoder.last_conv.8.weight"):
        assert k in keys, k


def test_output_stride_controls_dilation():
    m16 = build_model("deeplabv3plus_resnet50", 21, output_stride=16)
    m8 = build_model("deeplabv3plus_resnet50", 21, output_stride=8)
    assert m16.backbone.layer4[1].conv2.dilation != m8.backbone.layer4[1].conv2.dilation
    assert m16.aspp.aspp2.atrous_conv.dilation == (6, 6) and m8.aspp.aspp2.atrous_conv.dilation == (12, 12)


def test_batchnorm_matches_torch_and_fuses_residual_relu():
    torch.manual_seed(1)
    bn = ops.BatchNorm2d(7)
    ref = torch.nn.BatchNorm2d(7)
    x = torch.randn(4, 7, 6, 5, requires_grad=True)
    r = torch.randn(4, 7, 6, 5, requires_grad=True)
    x2, r2 = x.detach().clone().requires_grad_(True), r.detach().clone().requires_grad_(True)
    y = bn(x, relu=True, residual=r)
    y2 = torch.relu(ref(x2) + r2)
    assert torch.allclose(y, y2, atol=1e-5)
    g = torch.randn_like(y)
    y.backward(g)
    y2.backward(g)
    assert torch.allclose(x.grad, x2.grad, atol=1e-5) and torch.allclose(r.grad, r2.grad, atol=1e-6)
    assert torch.allclose(bn.weight.grad, ref.weight.grad, atol=1e-4)
    assert torch.allclose(bn.running_var, ref.running_var, atol=1e-5)
    bn.eval(), ref.eval()
    assert torch.allclose(bn(x.detach()), ref(x.detach()), atol=1e-5)


# --------------------------------------------------------------------------------------
# arena / optimiser / engine
# --------------------------------------------------------------------------------------
def test_arena_layout_buckets_and_views():
    m = build_model("fcn_resnet18", 21)
    ref = {k: v.clone() for k, v in m.state_dict().items()}
    arena = ParamArena([list(m.named_parameters())], bucket_bytes=1 << 20)
    # values survive the re-homing; views alias the flat buffers; conv weights are KRSC in memory
    for k, v in m.state_dict().items():
        assert torch.equal(v, ref[k]), k
    w = m.backbone.layer1[0].conv1.weight
    assert w.data.permute(0, 2, 3, 1).is_contiguous() and w.grad.data_ptr() != 0
    base = arena.flat_p.data_ptr()
    assert all(base <= s.param.data_ptr() < base + arena.total * 4 for s in arena.slots)
    # reverse registration order (last layer first) and contiguous, covering buckets
    names = [n for n, _ in m.named_parameters()]
    assert arena.slots[0].name == names[-1] and arena.slots[-1].name == names[0]
    assert arena.buckets[0].start == 0 and arena.buckets[-1].end == arena.total
    for a, b in zip(arena.buckets, arena.buckets[1:]):
        assert a.end == b.start
    assert len(arena.buckets) > 3
    assert all(s.offset % 64 == 0 for s in arena.slots)


def test_flat_sgd_matches_torch_sgd():
    """Same update rule as ``optim.SGD(momentum, weight_decay)`` (`train_pascal.py:118`), fed with
    identical gradients; ``grad_scale`` plays the role of DDP's 1/world average."""
    torch.manual_seed(0)
    m1 = build_model("fcn_resnet18", 4)
    m2 = build_model("fcn_resnet18", 4)
    m2.load_state_dict(m1.state_dict())
    ddp = DistributedDataParallel(m1)
    opt1 = FlatSGD(ddp.arena, 0.1, momentum=0.9, weight_decay=1e-3)
    opt2 = torch.optim.SGD(m2.parameters(), lr=0.1, momentum=0.9, weight_decay=1e-3)
    for it in range(3):
        g = torch.Generator().manual_seed(it)
        for p1, p2 in zip(m1.parameters(), m2.parameters()):
            gr = torch.randn(p2.shape, generator=g)
            p1.grad.copy_(gr * 2.0)          # "summed over 2 ranks"
            p2.grad = gr.clone()
        opt1.step(grad_scale=0.5, zero_grad=True)
        opt2.step()
    for (k, a), (_, b) in zip(m1.named_parameters(), m2.named_parameters()):
        assert torch.allclose(a, b, rtol=1e-5, atol=1e-6), k
    assert float(ddp.arena.flat_g.abs().max()) == 0.0        # step() zeroed the gradient arena


def test_engine_head_lr_groups_and_scheduler():
    m = build_model("fcn_resnet18", 21)
    eng = TrainEngine(m, ops.SoftmaxCrossEntropy(255), lr=0.01, head_lr_mult=10.0)
    assert [pg["lr"] for pg in eng.optimizer.param_groups] == [0.01, 0.1]
    sch = LR_Scheduler("poly", 0.01, 2, 10)
    sch(eng.optimizer, 5, 1)
    lr = 0.01 * (1 - 15 / 20) ** 0.9
    assert math.isclose(eng.optimizer.param_groups[0]["lr"], lr, rel_tol=1e-9)
    assert math.isclose(eng.optimizer.param_groups[1]["lr"], 10 * lr, rel_tol=1e-9)
    assert LR_Scheduler("const", 5e-8, 100, 7).lr_at(3, 50) == 5e-8       # the reference's effective schedule
    assert LR_Scheduler("cos", 1.0, 1, 10).lr_at(10, 0) == pytest.approx(0.0, abs=1e-12)
    assert LR_Scheduler("step", 1.0, 9, 10, lr_step=3).lr_at(0, 6) == pytest.approx(0.01)
    with pytest.raises(ValueError):
        LR_Scheduler("bogus", 1.0, 1, 1)


def test_checkpoint_reference_keyspace_roundtrip(tmp_path):
    """Weights file = what `train_pascal.py:230/:304` writes and `:103` reads: flat dict, ``module.``
    keys, fp32 CPU tensors, NCHW-contiguous; everything else lives in the ``.state`` sidecar."""
    torch.manual_seed(0)
    m = build_model("fcn_resnet18", 21)
    eng = TrainEngine(m, ops.SoftmaxCrossEntropy(255), lr=0.05)
    x = torch.randn(2, 3, 32, 32)
    y = torch.randint(0, 21, (2, 32, 32))
    eng.train_step(x, y)
    path = str(tmp_path / "models" / "net_epoch-0.pth")
    save_checkpoint(path, eng.ddp, eng.optimizer, {"epoch": 0, "best_val": 0.5})
    sd = torch.load(path, weights_only=True)
    assert isinstance(sd, dict) and all(k.startswith("module.") for k in sd)
    assert all(v.device.type == "cpu" and v.is_contiguous() for v in sd.values())
    assert all(v.dtype in (torch.float32, torch.int64) for v in sd.values())
    # a plain nn.DataParallel-wrapped copy of the architecture loads it strictly (the reference's line 103)
    plain = torch.nn.DataParallel(build_model("fcn_resnet18", 21))
    plain.load_state_dict(sd, strict=True)
    # and a file saved by such a wrapper (or un-prefixed, or {'state_dict': ...}) loads back into the arena
    for obj in (plain.state_dict(), plain.module.state_dict(), {"state_dict": plain.state_dict()}):
        p2 = str(tmp_path / "other.pth")
        torch.save(obj, p2)
        m3 = build_model("fcn_resnet18", 21)
        eng3 = TrainEngine(m3, ops.SoftmaxCrossEntropy(255), lr=0.05)
        load_weights(p2, eng3.ddp)
        for (k, a), (_, b) in zip(m.state_dict().items(), m3.state_dict().items()):
            assert torch.equal(a, b), k
        assert m3.backbone.conv1.weight.data_ptr() == eng3.ddp.arena.slots[-1].param.data_ptr()  # still in the arena
    # sidecar restores momentum / step count / extras
    m4 = build_model("fcn_resnet18", 21)
    eng4 = TrainEngine(m4, ops.SoftmaxCrossEntropy(255), lr=0.05)
    extra = load_checkpoint(path, eng4.ddp, eng4.optimizer)
    assert extra == {"epoch": 0, "best_val": 0.5, "rank_state": {}}     # one rank, nothing rank-specific saved
    assert torch.equal(eng4.ddp.arena.flat_m, eng.ddp.arena.flat_m)
    torch.manual_seed(5)            # dropout in the head
    l1 = eng.train_step(x, y)
    torch.manual_seed(5)
    l2 = eng4.train_step(x, y)
    assert torch.allclose(l1, l2, atol=1e-6)
    assert set(reference_state_dict(eng.ddp)) == set(sd)


def test_evaluate_miou_perfect_and_ignore():
    cm = ops.confusion_matrix(torch.tensor([[[0, 1], [2, 2]]]), torch.tensor([[[0, 1], [2, 255]]]), 3, 255)
    assert cm.tolist() == [[1, 0, 0], [0, 1, 0], [0, 0, 1]]
    miou, iou, acc = ops.miou_from_confusion(cm)
    assert miou == pytest.approx(1.0) and acc == pytest.approx(1.0)
    cm = ops.confusion_matrix(torch.tensor([[[0, 0], [1, 1]]]), torch.tensor([[[0, 1], [1, 1]]]), 3, 255)
    miou, iou, acc = ops.miou_from_confusion(cm)      # class 2 absent → excluded from the mean
    assert miou == pytest.approx((0.5 + 2 / 3) / 2) and acc == pytest.approx(0.75)


def test_softmax_ce_ignore_index_cpu():
    torch.manual_seed(0)
    logits = torch.randn(2, 5, 4, 4, requires_grad=True)
    tgt = torch.randint(0, 5, (2, 4, 4))
    tgt[0, 0] = 255
    loss = ops.softmax_cross_entropy(logits, tgt, 255)
    ref = torch.nn.functional.cross_entropy(logits, tgt, ignore_index=255)
    assert torch.allclose(loss, ref, atol=1e-6)
    all_ign = torch.full((2, 4, 4), 255)
    assert float(ops.softmax_cross_entropy(logits, all_ign, 255)) == 0.0   # no NaN on an all-ignored batch


def test_losses_named_by_the_reference():
    torch.manual_seed(0)
    out = torch.randn(2, 1, 8, 8)
    gt = (torch.rand(2, 1, 8, 8) > 0.7).float()
    l_b = class_balanced_cross_entropy_loss(out, gt, size_average=False, batch_average=True)
    # hand computation of the DEXTR balanced BCE
    pos, neg = gt, 1 - gt
    per = torch.nn.functional.binary_cross_entropy_with_logits(out, gt, reduction="none")
    want = (neg.sum() / gt.numel() * (pos * per).sum() + pos.sum() / gt.numel() * (neg * per).sum()) / 2
    assert torch.allclose(l_b, want, atol=1e-5)
    crit = SegmentationMultiLosses(1)
    three = (out, out * 0.5, out * 0.1)
    assert torch.allclose(crit(three, gt), sum(crit._one(o, gt) for o in three))
    crit21 = SegmentationMultiLosses(21)
    lg = torch.randn(2, 21, 8, 8)
    tg = torch.randint(0, 21, (2, 8, 8))
    assert torch.allclose(crit21(lg, tg), torch.nn.functional.cross_entropy(lg, tg), atol=1e-6)


def test_jsonl_logger_and_phase_timer(tmp_path):
    from distributedpytorch_amd.utils import PhaseTimer
    p = tmp_path / "log.jsonl"
    lg = JsonlLogger(str(p), echo=False)
    lg.log("train", epoch=1, loss=0.25)
    lg.close()
    rec = json.loads(p.read_text().splitlines()[0])
    assert rec["kind"] == "train" and rec["epoch"] == 1 and rec["loss"] == 0.25
    t = PhaseTimer(False)
    with t.phase("fwd"):
        sum(range(1000))
    s = t.summary()
    assert "fwd" in s and s["fwd"] >= 0.0


def test_gpu_augment_reference_path_cpu():
    """The fused crop/flip/rescale+normalise op (K8/K9) — CPU reference path: identity matrix with
    flip=0 reproduces (x/255-mean)/std and the label map; out-of-image label pixels become 255."""
    B, S = 2, 9
    img = torch.randint(0, 256, (B, S, S, 3), dtype=torch.uint8)
    lab = torch.randint(0, 21, (B, S, S), dtype=torch.uint8)
    mats = ops.make_affine_params(B, (S, S), (S, S), scales=(1.0, 1.0), flip_p=0.0, rng=np.random.default_rng(0))
    x, y = ops.affine_augment(img, lab, torch.from_numpy(mats), (S, S), out_dtype=torch.float32)
    mean = torch.tensor(ops.IMAGENET_MEAN).view(1, 3, 1, 1)
    std = torch.tensor(ops.IMAGENET_STD).view(1, 3, 1, 1)
    want = (img.permute(0, 3, 1, 2).float() - mean) / std       # mean/std are in 0..255 units
    assert x.shape == (B, 3, S, S) and torch.allclose(x, want, atol=1e-4)
    assert torch.equal(y.long(), lab.long())
    mats = ops.make_affine_params(B, (S, S), (S, S), scales=(0.5, 0.5), flip_p=0.0, rng=np.random.default_rng(0))
    _, y2 = ops.affine_augment(img, lab, torch.from_numpy(mats), (S, S), out_dtype=torch.float32)
    assert (y2 == 255).any()       # zoom-out exposes the border → ignore label


def test_loss_upsamples_lowres_logits_like_interpolate():
    """ops.softmax_cross_entropy with a label map larger than the logits = CE of the bilinearly
    (align_corners) upsampled logits — the contract of the fused GPU kernel, on the CPU path."""
    import torch.nn.functional as F
    from distributedpytorch_amd import ops
    torch.manual_seed(0)
    logits = torch.randn(2, 5, 9, 7, requires_grad=True)
    target = torch.randint(0, 5, (2, 33, 25))
    target[:, :3] = 255
    got = ops.softmax_cross_entropy(logits, target, 255)
    up = F.interpolate(logits, size=(33, 25), mode="bilinear", align_corners=True)
    ref = F.cross_entropy(up, target, ignore_index=255)
    assert torch.allclose(got, ref, atol=1e-6)
    g1, = torch.autograd.grad(got, logits)
    g2, = torch.autograd.grad(ref, logits)
    assert torch.allclose(g1, g2, atol=1e-6)


def test_engine_defers_final_upsample_only_in_training():
    from distributedpytorch_amd.models import build_model
    from distributedpytorch_amd.ops import SoftmaxCrossEntropy
    from distributedpytorch_amd.parallel import TrainEngine
    torch.manual_seed(0)
    model = build_model("deeplabv3plus_resnet18", 21)
    assert model.defer_upsample is False
    eng = TrainEngine(model, SoftmaxCrossEntropy(255), lr=0.01, device=torch.device("cpu"))
    assert model.defer_upsample is True
    x = torch.randn(2, 3, 65, 65)
    y = torch.randint(0, 21, (2, 65, 65))
    model.train()
    assert model(x).shape[-2:] == (17, 17)            # stride-4 logits, upsampled inside the loss
    model.eval()
    with torch.no_grad():
        assert model(x).shape[-2:] == (65, 65)        # evaluation sees full resolution
    l0 = float(eng.train_step(x, y))
    for _ in range(5):
        l1 = float(eng.train_step(x, y))
    assert l1 < l0


def test_conv_passthrough_routes_skip_gradient_cpu():
    """conv_bn_act(passthrough=True) returns the input as the block's identity; gradients equal the
    plain formulation (on the GPU the skip gradient is added inside the data-gradient kernel)."""
    from distributedpytorch_amd.ops import BatchNorm2d, Conv2d, conv_bn_act
    torch.manual_seed(1)
    conv, bn = Conv2d(8, 8, 3, padding=1, bias=False), BatchNorm2d(8)
    x1 = torch.randn(2, 8, 9, 9, requires_grad=True)
    x2 = x1.detach().clone().requires_grad_(True)
    out, ident = conv_bn_act(conv, bn, x1, relu=True, passthrough=True)
    (out + ident).pow(2).sum().backward()
    gw = conv.weight.grad.clone()
    conv.weight.grad = None
    (conv_bn_act(conv, bn, x2, relu=True) + x2).pow(2).sum().backward()
    assert torch.allclose(x1.grad, x2.grad, atol=1e-5)
    assert torch.allclose(gw, conv.weight.grad, atol=1e-5)
