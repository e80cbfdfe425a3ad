"""Harder checks of the hand-written kernels (VERDICT r1 "harden the tests"):

* headline problems of the packaged tile table (``ops/tuned_gfx950.json``: DeepLabv3+ R101 at 513x513) at batch 32
  through forward / data gradient / weight gradient WITH the table's tile configuration, judged by per-channel
  relative error and cosine similarity against an fp32 CPU convolution (not by one max-over-tensor number);
* an input beyond the second-generation kernels' 32-bit offsets (> 2 GiB) takes the first-generation kernels and
  gives the same numbers as the per-image launches;
* one training step of DeepLabv3+ Xception-65 (output stride 8) and of DeepLabv3+ R101 against the same model in
  fp32 on the CPU.

Run on an MI355X: ``pytest -m gpu``."""
import copy
import json
import math
import os

import pytest
import torch
import torch.nn.functional as F

from distributedpytorch_amd import ops
from distributedpytorch_amd.ops import _backend as B

pytestmark = pytest.mark.gpu
DEV = "cuda"
_TABLE = os.path.join(os.path.dirname(ops.__file__), "tuned_gfx950.json")


def _nhwc(t):
    return t.contiguous(memory_format=torch.channels_last)


def cosine(a, b):
    a, b = a.detach().double().flatten().cpu(), b.detach().double().flatten().cpu()
    return float((a @ b) / (a.norm() * b.norm()).clamp_min(1e-30))


def channel_rel(a, b, dim=1):
    """Worst per-channel max error relative to that channel's own largest reference magnitude (a quiet channel
    is not hidden behind a loud one); channels whose reference is ~0 everywhere are judged against the tensor."""
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    dims = [d for d in range(a.dim()) if d != dim]
    err = (a - b).abs().amax(dim=dims)
    mag = b.abs().amax(dim=dims).clamp_min(1e-2 * float(b.abs().max()) + 1e-30)
    return float((err / mag).max())


def channel_cos(a, b, dim=1):
    a = a.detach().double().cpu().movedim(dim, 0).flatten(1)
    b = b.detach().double().cpu().movedim(dim, 0).flatten(1)
    c = (a * b).sum(1) / (a.norm(dim=1) * b.norm(dim=1)).clamp_min(1e-30)
    live = b.norm(dim=1) > 1e-6 * b.norm()
    return float(c[live].min())


# (hi, cin, cout, k, stride, pad, dil, with_stats, check_wgrad): fwd problems of the table (its batch is 128)
HEADLINE = [
    (129, 64, 256, 1, 1, 0, 1, True, True),        # layer1 expand
    (129, 128, 128, 3, 2, 1, 1, True, True),       # layer2 strided 3x3 (the table sends it to the first generation)
    (65, 128, 128, 3, 1, 1, 1, True, True),        # layer2 3x3
    (33, 512, 512, 3, 1, 2, 2, True, False),       # layer4 atrous 3x3
    (33, 2048, 256, 3, 1, 12, 12, True, False),    # ASPP branch, K = 18432, taps outside the map
    (33, 1024, 2048, 1, 1, 0, 1, True, True),      # layer4 projection
    (129, 304, 256, 3, 1, 1, 1, True, False),      # decoder conv on the concat (Cin = 304)
]


def _table_cfg(n, hi, cin, ho, cout, k, stride, pad, dil, stats):
    with open(_TABLE) as f:
        table = json.load(f)
    key = json.dumps([n, hi, hi, cin, ho, ho, cout, k, k, stride, 1, [pad, pad], [dil, dil], False, stats])
    return table.get(key)


@pytest.mark.parametrize("case", HEADLINE)
def test_headline_problem_at_batch_32(case):
    from distributedpytorch_amd.ops import _hip_conv
    hi, cin, cout, k, stride, pad, dil, stats, check_wgrad = case
    n = 32
    ho = (hi + 2 * pad - dil * (k - 1) - 1) // stride + 1
    cfg = _table_cfg(128, hi, cin, ho, cout, k, stride, pad, dil, stats)
    assert cfg is not None, "problem missing from the packaged tile table"
    torch.manual_seed(hi * 7 + cin)
    # per-channel scales over two decades: a max-over-tensor tolerance would hide errors in the quiet channels
    x = torch.randn(n, cin, hi, hi) * (10 ** torch.linspace(-1, 1, cin)).view(1, cin, 1, 1)
    x = x.to(torch.bfloat16)
    w = (torch.randn(cout, cin, k, k) / math.sqrt(cin * k * k) *
         (10 ** torch.linspace(-1, 0.5, cout)).view(cout, 1, 1, 1)).to(torch.bfloat16)
    gy = torch.randn(n, cout, ho, ho).to(torch.bfloat16)
    sel = [0, n - 1]

    # ---- raw kernel with the table's tile configuration (forward + fused statistics) ----
    xg, wg = _nhwc(x.to(DEV)), _nhwc(w.to(DEV))
    st = torch.zeros(4, 2, cout, device=DEV) if stats else None
    y = _hip_conv.igemm(xg, wg, (ho, ho), stride, 1, (pad, pad), (dil, dil), None, st, cfg=cfg)
    xr = x[sel].float().requires_grad_(True)
    wr = w.float().requires_grad_(True)
    yr = F.conv2d(xr, wr, None, stride, pad, dil)
    assert cosine(y[sel], yr) > 0.99995
    assert channel_cos(y[sel], yr) > 0.9995
    assert channel_rel(y[sel], yr) < 2e-2
    if stats:
        s = st.sum(0)
        yf = y.float()
        assert channel_rel(s[0], yf.sum(dim=(0, 2, 3)), dim=0) < 2e-2
        assert channel_rel(s[1], (yf * yf).sum(dim=(0, 2, 3)), dim=0) < 2e-2

    # ---- autograd path: data gradient (two images against the CPU) and weight gradient (whole batch) ----
    xh = xg.clone().requires_grad_(True)
    wh = w.float().to(DEV).requires_grad_(True)
    yh = ops.conv2d(xh, wh, None, stride, pad, dil)
    yh.backward(_nhwc(gy.to(DEV)))
    yr.backward(gy[sel].float())
    assert cosine(xh.grad[sel], xr.grad) > 0.9999
    assert channel_cos(xh.grad[sel], xr.grad) > 0.999
    assert channel_rel(xh.grad[sel], xr.grad) < 2e-2
    if check_wgrad:
        wfull = w.float().requires_grad_(True)
        F.conv2d(x.float(), wfull, None, stride, pad, dil).backward(gy.float())
        assert cosine(wh.grad, wfull.grad) > 0.9999
        assert channel_cos(wh.grad, wfull.grad, dim=0) > 0.999       # per output channel
        assert channel_cos(wh.grad, wfull.grad, dim=1) > 0.999       # per input channel
        assert channel_rel(wh.grad, wfull.grad, dim=0) < 2e-2


def test_conv_beyond_2gib_runs_as_batch_chunks():
    """8 x 128 x 1025 x 1025 bf16 = 2.15 GB: past the 32-bit byte offsets of the second-generation kernels
    (dp_conv_igemm2 returns -4 for the whole batch; ops/_hip_conv.py cuts it into image chunks on the same kernels).
    The whole-batch results must equal the per-image launches (269 MB each) and an fp32 CPU convolution of
    sampled rows."""
    from distributedpytorch_amd.ops import _hip_conv
    n, cin, cout, hw = 8, 128, 16, 1025
    torch.manual_seed(2)
    x = _nhwc(torch.randn(n, cin, hw, hw, device=DEV, dtype=torch.bfloat16))
    assert x.numel() * 2 >= (1 << 31) - (1 << 24)
    for k, pad in ((1, 0), (3, 1)):
        w = _nhwc((torch.randn(cout, cin, k, k, device=DEV) / math.sqrt(cin * k * k)).to(torch.bfloat16))
        lib = B.lib()
        from distributedpytorch_amd.ops._hip_conv import _lib
        y_probe = torch.empty(n, cout, hw, hw, device=DEV, dtype=torch.bfloat16).contiguous(memory_format=torch.channels_last)
        rc = _lib().dp_conv_igemm2(B.ptr(x), B.ptr(w), B.ptr(y_probe), None, None, 1, n, hw, hw, cin, hw, hw, cout, k, k,
                                   1, 1, pad, pad, 1, 1, cout, 1, B.stream_ptr())
        assert rc == -4, f"second-generation kernel accepted a {x.numel() * 2 / 2**30:.2f} GiB input (rc {rc})"
        assert 1 <= _hip_conv._images_per_launch(n, x) < n
        xh = x.detach().requires_grad_(True)
        wh = w.float().requires_grad_(True)
        y = ops.conv2d(xh, wh, None, 1, pad, 1)
        gy = _nhwc(torch.randn_like(y))
        y.backward(gy)
        # per-image launches
        dw_sum = torch.zeros_like(wh)
        for i in (0, n - 1):
            xi = x[i:i + 1].detach().clone(memory_format=torch.channels_last).requires_grad_(True)
            wi = w.float().requires_grad_(True)
            yi = ops.conv2d(xi, wi, None, 1, pad, 1)
            yi.backward(gy[i:i + 1].clone(memory_format=torch.channels_last))
            assert channel_rel(y[i:i + 1], yi) < 1e-2 and cosine(y[i:i + 1], yi) > 0.99999
            assert channel_rel(xh.grad[i:i + 1], xi.grad) < 1e-2 and cosine(xh.grad[i:i + 1], xi.grad) > 0.99999
        wall = torch.zeros_like(wh)
        for i in range(n):
            wi = w.float().requires_grad_(True)
            xi = x[i:i + 1].detach().clone(memory_format=torch.channels_last)
            ops.conv2d(xi, wi, None, 1, pad, 1).backward(gy[i:i + 1].clone(memory_format=torch.channels_last))
            wall += wi.grad
        assert cosine(wh.grad, wall) > 0.9999 and channel_rel(wh.grad, wall, dim=0) < 1e-2
        # fp32 CPU convolution of a strip of the last image
        rows = slice(500, 520)
        xs = x[n - 1:n, :, rows.start - pad:rows.stop + pad].float().cpu()
        ys = F.conv2d(xs, w.float().cpu(), None, 1, (0, pad), 1)
        assert channel_rel(y[n - 1:n, :, rows], ys) < 1.5e-2
        del xh, y, gy, y_probe
        torch.cuda.empty_cache()


@pytest.mark.parametrize("name,kwargs,size,gain,floor,margin", [
    ("deeplabv3plus_xception65", dict(output_stride=8), 129, 0.08, 0.75, 0.12),
    ("deeplabv3plus_resnet101", dict(), 257, 0.15, 0.65, 0.15),
])
def test_model_training_step_matches_fp32_cpu(name, kwargs, size, gain, floor, margin):
    """One training step (batch statistics) on the kernels against the same model in fp32 on the CPU's torch path:
    loss and the gradients of every convolution.

    A random-init residual stack at full gain is chaotic — rounding only the INPUT to bf16 moves the fp32 CPU
    gradients to a median cosine of 0.42 (X65) / 0.04 (R101) against themselves (tools/diag_model_step.py,
    gpurun_out/q/diag_*.log) — so the BatchNorm closing each residual branch gets gamma x 0.15 (a network near the
    identity, as under zero-init-residual).  There the fp32 noise floor is 0.98, the bf16 kernels land at 0.73-0.87
    with OR without the fused paths, and a wiring error in a fused path shows as fused << plain.

    Xception-65 (20 residual blocks, 768-wide stored flow) at x 0.15 spreads over 0.44-0.90 from run to run for
    EITHER variant and EITHER of two library versions (gpurun_out/bh: fp32 atomics sum in launch order); at x 0.08 /
    x 0.04 nine runs land in 0.878-0.934 (gpurun_out/bi), while x 0.06 sits at 0.69 / 0.71 (the level depends on the draw more than on the gain): that model runs at the
    measured x 0.08 with a floor of 0.75."""
    from distributedpytorch_amd.models import build_model
    from distributedpytorch_amd.ops import SoftmaxCrossEntropy, _hip_conv, sepconv
    from distributedpytorch_amd.parallel import TrainEngine
    torch.manual_seed(0)
    m_cpu = build_model(name, 21, **kwargs).train()
    with torch.no_grad():
        for m in m_cpu.modules():
            if hasattr(m, "running_mean"):
                m.weight.uniform_(0.7, 1.3); m.bias.normal_(0, 0.3)
        for k, p in m_cpu.named_parameters():
            if k.endswith("bn3.weight") or (".rep." in k and k.endswith(".bn.weight") and not k.endswith("sep.bn.weight")):
                p.mul_(gain)
    if hasattr(getattr(m_cpu, "backbone", None), "zero_pad_channels"):
        m_cpu.backbone.zero_pad_channels()      # Xception's padding channels: zero again after the re-initialisation
    for m in m_cpu.modules():            # dropout off: the fused counter-based generator is not torch's
        if isinstance(getattr(m, "p_drop", None), tuple):
            m.p_drop = (0.0, 0.0)
        elif hasattr(m, "p_drop"):
            m.p_drop = 0.0
    g = torch.Generator().manual_seed(1)
    x = torch.randn(4, 3, size, size, generator=g)
    lab = torch.randint(0, 21, (4, 8, 8), generator=g)
    lab = F.interpolate(lab[:, None].float(), size=(size, size), mode="nearest")[:, 0].long()
    lab[:, :4] = 255
    crit = SoftmaxCrossEntropy(255)
    got = {}
    for var, fold, prologue in (("fused", True, True), ("plain", False, False)):
        old = sepconv._FOLD, _hip_conv._BN_PROLOGUE, B.set_deterministic(True)    # reproducible: no atomics order
        sepconv._FOLD, _hip_conv._BN_PROLOGUE = fold, prologue
        try:
            eng = TrainEngine(copy.deepcopy(m_cpu), crit, lr=0.0, momentum=0.0, weight_decay=0.0,
                              device=torch.device("cuda", 0))
            loss = crit(eng.ddp(eng.prepare_input(x)), lab.cuda())
            loss.backward()
            torch.cuda.synchronize()
            got[var] = (float(loss.detach()), {s.name: s.param.grad.detach().float().cpu().clone()
                                               for s in eng.ddp.arena.slots})
        finally:
            sepconv._FOLD, _hip_conv._BN_PROLOGUE = old[:2]
            B.set_deterministic(old[2])
    l_cpu = crit(m_cpu(x), lab)
    l_cpu.backward()
    ref = {k: p.grad.detach().float() for k, p in m_cpu.named_parameters() if p.grad is not None}
    names = [k for k in ref if ref[k].dim() == 4 and float(ref[k].norm()) > 1e-7]

    def summary(a):
        cs = [cosine(a[k], ref[k]) for k in names]
        return cs[-1], sorted(cs)[len(cs) // 2]

    (lf, gf), (lp, gp) = got["fused"], got["plain"]
    last_f, med_f = summary(gf)
    last_p, med_p = summary(gp)
    print(name, "loss cpu/fused/plain", float(l_cpu.detach()), lf, lp, "last-layer cos", last_f, last_p, "median cos", med_f, med_p)
    assert abs(lf - float(l_cpu)) < 0.02 * abs(float(l_cpu))
    # The median moves from run to run of ONE binary (fp32 atomics sum the weight gradients in launch order; a sum
    # that rounds differently flips ReLU masks downstream): 0.73-0.86 for either variant over seven runs
    # (gpurun_out/al, am) — so the floor and the fused-vs-plain margin are set from that spread; a wiring error
    # in a fused path lands far below both (0.04-0.4).
    assert last_f > 0.995 and med_f > floor, (last_f, med_f)
    assert med_f > med_p - margin, (med_f, med_p)


def test_xception_padded_flow_stays_the_728_channel_network(tmp_path):
    """Aligned Xception-65 stores its 728-channel flow 768 wide on the HIP backend (models/xception.py): after real
    optimizer steps on the kernels (momentum, weight decay, fused BN folds, atomics) the padding weights are still
    exact zeros, and the checkpoint has the architecture's shapes and loads into an unpadded model."""
    from distributedpytorch_amd.models import build_model
    from distributedpytorch_amd.models.xception import MID_CHANNELS
    from distributedpytorch_amd.ops import SoftmaxCrossEntropy
    from distributedpytorch_amd.parallel import TrainEngine
    from distributedpytorch_amd.utils import checkpoint
    torch.manual_seed(0)
    net = build_model("deeplabv3plus_xception65", 21, output_stride=8)
    bb = net.backbone
    assert bb.mid_physical == 768, "padding is the default on the HIP backend"
    eng = TrainEngine(net, SoftmaxCrossEntropy(255), lr=0.01, momentum=0.9, weight_decay=5e-4,
                      device=torch.device("cuda", 0))
    g = torch.Generator().manual_seed(1)
    for _ in range(3):
        x = torch.randn(4, 3, 129, 129, generator=g)
        lab = torch.randint(0, 21, (4, 129, 129), generator=g)
        loss = eng.train_step(x.cuda(), lab.cuda())
    assert torch.isfinite(loss).item()
    n = 0
    for name, t, dims in bb._padded_tensors():
        if t.dim() == 4 or name.endswith(".bias"):
            for d in dims:
                assert float(t.detach().narrow(d, MID_CHANNELS, 40).float().abs().max()) == 0.0, name
                n += 1
    assert n > 150
    path = str(tmp_path / "x65.pth")
    checkpoint.save_checkpoint(path, eng.model, sync=False)
    sd = torch.load(path, weights_only=True)
    assert sd["module.backbone.middle.0.rep.0.sep.pointwise.weight"].shape == (728, 728, 1, 1)
    assert sd["module.backbone.block20.skip.weight"].shape == (1024, 728, 1, 1)
    os.environ["DPAMD_X65_PAD"] = "0"
    try:
        plain = build_model("deeplabv3plus_xception65", 21, output_stride=8)
    finally:
        del os.environ["DPAMD_X65_PAD"]
    assert plain.backbone.mid_physical == 728
    res = checkpoint.load_weights(path, plain)
    assert not res.missing_keys and not res.unexpected_keys


def _randomise_bn(net):
    """BatchNorm shifts / scales away from (0, 1).  With shift 0 a scale whose output only meets ReLU -> depthwise
    convolution -> BatchNorm is scale-invariant: its TRUE gradient is ~0 (fp64: 0.7-0.9 next to neighbours of 250-500)
    and what bf16 kernels compute for it is rounding noise that no two launches agree on
    (profiles/r3_nondeterminism_xception_default_init.md).  Off that degenerate point every gradient of the stack is
    well-conditioned and can be held to tight tolerances."""
    with torch.no_grad():
        for m in net.modules():
            if hasattr(m, "running_mean"):
                m.weight.uniform_(0.6, 1.4); m.bias.normal_(0, 0.5)
    return net


def _three_runs(net, x, monkeypatch, chunk_limit):
    """(y, dx, parameter gradients) of one launch, the same launch again, and image chunks."""
    from distributedpytorch_amd.ops import _hip_conv
    gy, res = None, {}
    big = _hip_conv._OFFSET_LIMIT
    for tag, limit in (("one", big), ("again", big), ("chunks", chunk_limit)):
        monkeypatch.setattr(_hip_conv, "_OFFSET_LIMIT", limit)
        net.zero_grad(set_to_none=True)
        xi = x.detach().clone(memory_format=torch.channels_last).requires_grad_(True)
        y = net(xi)
        if gy is None:
            gy = _nhwc(torch.randn_like(y))
        y.backward(gy)
        torch.cuda.synchronize()
        res[tag] = (y.detach().float(), xi.grad.float(), [p.grad.detach().float().clone() for p in net.parameters()])
    monkeypatch.setattr(_hip_conv, "_OFFSET_LIMIT", big)
    return res


def _check_runs(res, det: bool):
    (y1, dx1, g1), (yr, dxr, gr), (y2, dx2, g2) = res["one"], res["again"], res["chunks"]
    if det:
        # deterministic mode: no address has two concurrent writers -> the same launch twice is BIT-identical
        assert torch.equal(y1, yr) and torch.equal(dx1, dxr)
        for a, r in zip(g1, gr):
            assert torch.equal(a, r)
    # chunks regroup the partial sums: an fp32 statistic moves in its last bit, a handful of bf16 roundings / ReLU
    # masks downstream flip with it (each worth ~1e-3 of a gradient's norm).  Measured on these stacks
    # (profiles/r3_nondeterminism_summary.md): >= 0.9989 deterministic, >= 0.9989 default, folded path
    floor = 0.9985 if det else 0.997
    assert cosine(y1, y2) > 0.99999 and cosine(dx1, dx2) > floor
    for a, b in zip(g1, g2):
        if float(a.norm()) > 1e-6:
            assert cosine(a, b) > floor, (cosine(a, b), tuple(a.shape))


class _DenseStack(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.c1 = ops.Conv2d(32, 64, 3, padding=1, bias=False); self.b1 = ops.BatchNorm2d(64)
        self.c2 = ops.Conv2d(64, 64, 1, bias=False); self.b2 = ops.BatchNorm2d(64)
        self.c3 = ops.Conv2d(64, 32, 3, padding=2, dilation=2, bias=False); self.b3 = ops.BatchNorm2d(32)

    def forward(self, x):
        a = ops.conv_bn_act(self.c1, self.b1, x, relu=True)
        b = ops.conv_bn_act(self.c2, self.b2, a, relu=True, residual=a)
        return ops.conv_bn_act(self.c3, self.b3, b, relu=True)


@pytest.mark.parametrize("det", [True, False])
def test_batch_chunks_equal_one_launch(monkeypatch, det):
    """The image-chunk path of the convolution wrappers (taken beyond 2 GiB) forced on a small conv-BN-ReLU stack:
    forward, fused BN-backward data gradients (with the 1-bit mask of a residual output) and weight gradients.
    Deterministic mode: two identical launches are bit-identical and chunks equal one launch to fp32 regrouping;
    default mode (float atomics): the same to a looser bound."""
    from distributedpytorch_amd.ops import _hip_conv
    torch.manual_seed(3)
    net = _randomise_bn(_DenseStack()).to(DEV).train()
    x = _nhwc(torch.randn(6, 32, 33, 33, device=DEV).to(torch.bfloat16))
    with B.deterministic_mode(det):
        res = _three_runs(net, x, monkeypatch, 2 * 32 * 33 * 33 * 2 + 1)
        monkeypatch.setattr(_hip_conv, "_OFFSET_LIMIT", 2 * 32 * 33 * 33 * 2 + 1)
        assert _hip_conv._images_per_launch(6, x) == 2          # (the patched limit: two images per launch)
    _check_runs(res, det)


@pytest.mark.parametrize("det", [True, False])
def test_depthwise_batch_chunks_equal_one_launch(monkeypatch, det):
    """The same for the depthwise path: two Xception blocks (rolling-window depthwise kernels with fused statistics,
    folded separable units, the fused BN-backward data gradient with shortcut add) in image chunks."""
    from distributedpytorch_amd.models.xception import XceptionBlock
    torch.manual_seed(4)
    blk = torch.nn.Sequential(XceptionBlock(64, 64, 3, dilation=2), XceptionBlock(64, 128, 2, stride=1, dilation=1))
    blk = _randomise_bn(blk).to(DEV).train()
    x = _nhwc(torch.randn(6, 64, 33, 33, device=DEV).to(torch.bfloat16))
    with B.deterministic_mode(det):
        res = _three_runs(blk, x, monkeypatch, 2 * 64 * 33 * 33 * 2 + 1)
    _check_runs(res, det)


@pytest.mark.parametrize("name,kwargs,size", [
    ("deeplabv3plus_resnet101", dict(), 129),
    ("deeplabv3plus_xception65", dict(output_stride=8), 129),
    ("danet_resnet50", dict(), 128),
])
def test_deterministic_mode_training_step_is_bit_identical(name, kwargs, size):
    """DPAMD_DETERMINISTIC / ops.set_deterministic: two training steps from the same state through the WHOLE path
    (engine, arena gradient sinks, fused loss, SGD) give bit-identical losses, gradients and updated weights."""
    from distributedpytorch_amd.models import build_model
    from distributedpytorch_amd.ops import SoftmaxCrossEntropy, batchnorm
    from distributedpytorch_amd.parallel import TrainEngine
    from distributedpytorch_amd.utils.losses import SegmentationMultiLosses
    torch.manual_seed(0)
    base = build_model(name, 21, **kwargs)
    g = torch.Generator().manual_seed(1)
    x = torch.randn(4, 3, size, size, generator=g)
    lab = torch.randint(0, 21, (4, size, size), generator=g)
    lab[:, :3] = 255
    out = []
    with B.deterministic_mode(True):
        for _ in range(2):
            crit = SegmentationMultiLosses(21) if name.startswith("danet") else SoftmaxCrossEntropy(255)
            eng = TrainEngine(copy.deepcopy(base), crit, lr=0.01, momentum=0.9, weight_decay=1e-4,
                              device=torch.device("cuda", 0))
            torch.manual_seed(7)                     # F.dropout of the DANet classifiers; the fused dropout's keys
            batchnorm._drop_calls = 0                # come from torch's seed and a per-process call counter
            losses = [eng.train_step(x.cuda(), lab.cuda()).detach().float().cpu().clone() for _ in range(2)]
            torch.cuda.synchronize()
            out.append((losses, [p.detach().float().cpu().clone() for p in eng.model.parameters()]))
    (l1, p1), (l2, p2) = out
    assert all(torch.equal(a, b) for a, b in zip(l1, l2)), (l1, l2)
    assert all(torch.isfinite(a).all() for a in l1)
    bad = [i for i, (a, b) in enumerate(zip(p1, p2)) if not torch.equal(a, b)]
    assert not bad, f"{len(bad)} of {len(p1)} parameters differ between two deterministic runs"


@pytest.mark.parametrize("stride", [1, 2])
def test_bottleneck_tail_fold_matches_fp32_cpu(stride):
    """ops.bnfold: the last BatchNorm of a bottleneck folded into conv3's gradients (T = dm^T a2, A = a2^T a2,
    G = W^T diag(kb) W; the pre-BN tensor is neither kept nor read in backward).  Three bottlenecks (the first with a
    projection shortcut) against the same modules in fp32 on the CPU, and against the unfolded kernels."""
    from distributedpytorch_amd.models.resnet import Bottleneck, _Projection
    from distributedpytorch_amd.ops import bnfold
    torch.manual_seed(11)

    class Net(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.stem = ops.Conv2d(16, 64, 3, padding=1, bias=False); self.sbn = ops.BatchNorm2d(64)
            self.b0 = Bottleneck(64, 32, stride, 1, _Projection(64, 128, stride))
            self.b1 = Bottleneck(128, 32, 1, 2)
            self.b2 = Bottleneck(128, 32, 1, 1)
            self.b1.input_exclusive = self.b2.input_exclusive = True
            self.b0.fold_tail = self.b1.fold_tail = True        # b2's output goes to the loss: not folded
            self.head = ops.Conv2d(128, 8, 1)

        def forward(self, x):
            x = ops.conv_bn_act(self.stem, self.sbn, x, relu=True)
            return self.head(self.b2(self.b1(self.b0(x))))

    net = _randomise_bn(Net()).train()
    x = torch.randn(4, 16, 33, 31).to(torch.bfloat16)
    ref = copy.deepcopy(net).float()
    with torch.no_grad():
        for p_ in ref.parameters():
            if p_.dim() == 4:
                p_.copy_(p_.to(torch.bfloat16).float())
    yr = ref(x.float())
    gy = torch.randn_like(yr).to(torch.bfloat16)
    yr.backward(gy.float())
    gref = {k: p_.grad for k, p_ in ref.named_parameters()}
    res = {}
    bnfold.COUNTS.update(folded=0, fallback=0)
    for tag, on in (("fold", True), ("plain", False)):
        old = bnfold._ENABLED, B.set_deterministic(True)       # the same numbers on every run of one binary
        bnfold._ENABLED = on
        try:
            g = copy.deepcopy(net).to(DEV)
            xi = _nhwc(x.to(DEV)).requires_grad_(True)
            y = g(xi)
            y.backward(_nhwc(gy.to(DEV)))
            torch.cuda.synchronize()
            res[tag] = (y.detach().float().cpu(), xi.grad.float().cpu(),
                        {k: p_.grad.detach().float().cpu() for k, p_ in g.named_parameters()},
                        {k: b.detach().float().cpu() for k, b in g.named_buffers()})
        finally:
            bnfold._ENABLED = old[0]
            B.set_deterministic(old[1])
    assert bnfold.COUNTS == {"folded": 2, "fallback": 0}, bnfold.COUNTS
    yf, dxf, gf, bf_ = res["fold"]
    yp, dxp, gp, bp = res["plain"]
    assert cosine(yf, yp) > 0.99999                         # the forward kernels are the same
    for k in bf_:
        assert torch.allclose(bf_[k], bp[k], rtol=1e-4, atol=1e-3), k       # running statistics
    assert cosine(yf, yr) > 0.9995
    worst = {}
    for k in gref:
        cf, cp = cosine(gf[k], gref[k]), cosine(gp[k], gref[k])
        worst[k] = (cf, cp)
        # the folded gradients are as close to fp32 as the unfolded ones
        assert cf > 0.97 and cf > cp - 0.01, (k, cf, cp)
        rf = float((gf[k] - gref[k]).norm() / gref[k].norm().clamp_min(1e-12))
        rp = float((gp[k] - gref[k]).norm() / gref[k].norm().clamp_min(1e-12))
        assert rf < max(1.5 * rp, rp + 0.02), (k, rf, rp)
    assert cosine(dxf, dxp) > 0.995


def _no_dropout(model):
    for m in model.modules():
        if isinstance(getattr(m, "p_drop", None), tuple):
            m.p_drop = (0.0, 0.0)
        elif hasattr(m, "p_drop"):
            m.p_drop = 0.0
    return model


@pytest.mark.parametrize("name,kwargs", [("deeplabv3plus_resnet50", dict()),
                                         ("deeplabv3plus_xception65", dict(output_stride=16))])
def test_graph_step_replays_train_exactly_like_eager_steps(name, kwargs):
    """TrainEngine(graph_step=True): the whole step replayed from one hipGraph.  Deterministic mode, dropout off: eight
    steps with a moving learning rate (two parameter groups at 1x / 10x) and a new batch each time give BIT-identical
    losses and weights to the eager engine — the inputs reach the replay through static buffers, the learning rate through
    the device scalar, and the capture itself trains nothing."""
    from distributedpytorch_amd.models import build_model
    from distributedpytorch_amd.ops import SoftmaxCrossEntropy
    from distributedpytorch_amd.parallel import TrainEngine
    torch.manual_seed(0)
    base = _no_dropout(build_model(name, 21, **kwargs))
    g = torch.Generator().manual_seed(1)
    xs = [torch.randn(4, 3, 97, 97, generator=g) for _ in range(8)]
    ys = [torch.randint(0, 21, (4, 97, 97), generator=g, dtype=torch.uint8) for _ in range(8)]
    out = {}
    with B.deterministic_mode(True):
        for mode in ("eager", "graph"):
            eng = TrainEngine(copy.deepcopy(base), SoftmaxCrossEntropy(255), lr=0.01, momentum=0.9, weight_decay=1e-4,
                              head_lr_mult=10.0, device=torch.device("cuda", 0), graph_step=mode == "graph")
            losses = []
            for i, (x, y) in enumerate(zip(xs, ys)):
                for pg, mult in zip(eng.optimizer.param_groups, (1.0, 10.0)):
                    pg["lr"] = 0.01 * mult * (1.0 - i / 10.0) ** 0.9
                losses.append(eng.train_step(x.cuda(), y.cuda()).detach().float().cpu().clone())
            torch.cuda.synchronize()
            out[mode] = (losses, [p.detach().float().cpu().clone() for p in eng.model.parameters()],
                         eng.global_step, eng.optimizer._steps, len(eng._graphs))
    (l1, p1, gs1, os1, ng1), (l2, p2, gs2, os2, ng2) = out["eager"], out["graph"]
    assert ng1 == 0 and ng2 == 1 and gs1 == gs2 == 8 and os1 == os2 == 8
    assert all(torch.isfinite(a).all() for a in l1)
    assert all(torch.equal(a, b) for a, b in zip(l1, l2)), ([float(a) for a in l1], [float(a) for a in l2])
    bad = [i for i, (a, b) in enumerate(zip(p1, p2)) if not torch.equal(a, b)]
    assert not bad, f"{len(bad)} of {len(p1)} parameters differ between eager and replayed steps"


def test_graph_step_draws_new_dropout_masks_per_replay_and_runs_the_aux_stream():
    """Default mode (weight gradients on the auxiliary stream inside the capture), dropout on: the same batch replayed
    at learning rate 0 gives a different loss every time — the fused dropouts mix the device counter the captured step
    increments into their keys — and the same weights; with dropout off the losses repeat exactly."""
    from distributedpytorch_amd.models import build_model
    from distributedpytorch_amd.ops import SoftmaxCrossEntropy
    from distributedpytorch_amd.parallel import TrainEngine
    torch.manual_seed(0)
    base = build_model("deeplabv3plus_resnet50", 21)
    x = torch.randn(4, 3, 97, 97).cuda()
    y = torch.randint(0, 21, (4, 97, 97), dtype=torch.uint8).cuda()
    for drop in (True, False):
        model = copy.deepcopy(base) if drop else _no_dropout(copy.deepcopy(base))
        with B.deterministic_mode(not drop):          # (dropout off: deterministic, so that equal means bit-equal)
            eng = TrainEngine(model, SoftmaxCrossEntropy(255), lr=0.0, momentum=0.0, weight_decay=0.0,
                              device=torch.device("cuda", 0), graph_step=True)
            w0 = [p.detach().clone() for p in eng.model.parameters()]
            losses = [float(eng.train_step(x, y)) for _ in range(9)]
        assert len(eng._graphs) == 1
        replayed = losses[4:]                         # steps 0-2 eager, step 3 = capture + first replay
        assert all(math.isfinite(v) for v in replayed)
        if drop:
            assert len({round(v, 6) for v in replayed}) == len(replayed), replayed
            assert int(next(iter(eng._graphs.values()))["drop_word"].item()) == 6
        else:
            assert max(replayed) == min(replayed), replayed
        # learning rate 0: only the BatchNorm running statistics move, the parameters do not
        assert all(torch.equal(a, b.detach()) for a, b in zip(w0, eng.model.parameters()))
