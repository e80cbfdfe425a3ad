"""Numerics of every HIP kernel against a plain PyTorch fp32 reference of the same op
(SURVEY.md §4 item 2).  Run on an MI355X: ``pytest -m gpu``.

Inputs are rounded to bf16 first so the comparison isolates the kernel's arithmetic (fp32
accumulation, bf16 output rounding) from input quantisation."""
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from distributedpytorch_amd import ops
from distributedpytorch_amd.ops import _backend as B

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _bf(t):
    return t.to(torch.bfloat16)


def _nhwc(t):
    return t.contiguous(memory_format=torch.channels_last)


def _rel(a, b):
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-6))


def test_library_loaded():
    assert B.lib_built(), "libdpamd_hip.so missing: run python -m distributedpytorch_amd.csrc.build"
    B.lib()


# ----------------------------------------------------------------------------- conv
CONV_CASES = [
    # n, cin, cout, h, w, k, stride, pad, dil
    (2, 64, 128, 33, 33, 1, 1, 0, 1),
    (2, 64, 64, 17, 19, 3, 1, 1, 1),
    (1, 128, 256, 33, 33, 3, 1, 2, 2),
    (1, 256, 256, 33, 33, 3, 1, 6, 6),
    (1, 64, 128, 33, 33, 3, 1, 12, 12),
    (1, 64, 64, 33, 33, 3, 1, 18, 18),
    (2, 64, 128, 65, 65, 3, 2, 1, 1),
    (2, 128, 256, 33, 35, 1, 2, 0, 1),
    (2, 304, 256, 17, 17, 3, 1, 1, 1),     # Cin not a multiple of 64 (decoder concat)
    (2, 256, 48, 19, 19, 1, 1, 0, 1),      # Cout = 48
    (2, 256, 21, 19, 19, 1, 1, 0, 1),      # classifier: Cout = 21 (padded internally)
    (2, 3, 64, 65, 65, 7, 2, 3, 1),        # RGB stem
    (3, 2048, 256, 9, 9, 3, 1, 6, 6),      # deep K, halo larger than the map
]


@pytest.mark.parametrize("case", CONV_CASES)
def test_conv_fwd_bwd(case):
    n, cin, cout, h, w, k, stride, pad, dil = case
    torch.manual_seed(0)
    x = _bf(torch.randn(n, cin, h, w, device=DEV))
    wt = _bf(torch.randn(cout, cin, k, k, device=DEV) / math.sqrt(cin * k * k))
    bias = torch.randn(cout, device=DEV) if cout == 21 else None
    # reference on the CPU in fp32: independent of MIOpen
    xr = x.float().cpu().requires_grad_(True)
    wr = wt.float().cpu().requires_grad_(True)
    yr = F.conv2d(xr, wr, None if bias is None else bias.cpu(), stride, pad, dil)
    gy = _bf(torch.randn_like(yr)).to(DEV)
    yr.backward(gy.float().cpu())

    xh = _nhwc(x).requires_grad_(True)
    wh = wt.float().requires_grad_(True)
    yh = ops.conv2d(xh, wh, bias, stride, pad, dil)
    assert yh.shape == yr.shape and yh.dtype == torch.bfloat16
    assert _rel(yh, yr) < 1.5e-2
    yh.backward(_nhwc(gy))
    if cin != 3:
        assert _rel(xh.grad, xr.grad) < 1.5e-2
    assert _rel(wh.grad, wr.grad) < 1.5e-2


@pytest.mark.parametrize("case", [
    # n, cin, cout, h, w, k, stride, pad
    (2, 3, 64, 65, 65, 7, 2, 3),
    (1, 3, 64, 513, 513, 7, 2, 3),        # the headline stem
    (3, 4, 64, 66, 40, 7, 2, 3),          # RGB + guidance channel (the reference's nInputChannels = 4), even sizes
    (2, 3, 32, 33, 35, 3, 2, 1),          # Xception entry conv
    (2, 1, 16, 20, 31, 5, 1, 2),
])
def test_stem_window_conv(case):
    """Small-Cin stem as row-segment windows (csrc/conv_igemm2.hip dp_conv_igemm2w, conv_wgrad.hip
    dp_conv_wgrad_win) against an fp32 CPU convolution: output, fused BN statistics, weight gradient."""
    from distributedpytorch_amd.ops import _hip_conv
    n, cin, cout, h, w, k, stride, pad = case
    torch.manual_seed(0)
    x = _bf(torch.randn(n, cin, h, w, device=DEV))
    wt = _bf(torch.randn(cout, cin, k, k, device=DEV) / math.sqrt(cin * k * k))
    xr = x.float().cpu()
    wr = wt.float().cpu().requires_grad_(True)
    yr = F.conv2d(xr, wr, None, stride, pad)
    gy = _bf(torch.randn_like(yr)).to(DEV)
    yr.backward(gy.float().cpu())

    wh = wt.float().requires_grad_(True)
    assert _hip_conv.stem_window_ok(x, wh, (stride, stride), (pad, pad), (1, 1))
    yh, stats = ops.conv2d(_nhwc(x), wh, None, stride, pad, 1, with_stats=True)
    assert yh.shape == yr.shape and yh.dtype == torch.bfloat16
    assert _rel(yh, yr) < 1.5e-2
    st = stats.sum(0) if stats.dim() == 3 else stats
    ref1 = yr.detach().sum(dim=(0, 2, 3))
    ref2 = (yr.detach() ** 2).sum(dim=(0, 2, 3))
    assert float((st[0].cpu() - ref1).abs().max()) < 2e-2 * float(ref2.sqrt().max()) + 1e-2 * float(ref1.abs().max())
    assert _rel(st[1], ref2) < 2e-2
    yh.backward(_nhwc(gy))
    assert wh.grad.shape == wr.grad.shape
    assert _rel(wh.grad, wr.grad) < 1.5e-2


def test_conv_identity_asymmetric():
    """A = I check with an asymmetric B: catches swapped row/col maps of the MFMA C layout."""
    cin = cout = 64
    x = torch.zeros(1, cin, 8, 16, device=DEV)
    for i in range(cin):
        x[0, i, i // 16, i % 16] = 1.0   # pixel p has a one in channel p (p < 64)
    w = torch.arange(cout * cin, device=DEV, dtype=torch.float32).view(cout, cin, 1, 1) % 251 / 64.0
    y = ops.conv2d(_nhwc(_bf(x)), _bf(w).float(), None, 1, 0, 1)
    ref = F.conv2d(_bf(x).float().cpu(), _bf(w).float().cpu())
    assert torch.allclose(y.float().cpu(), ref, atol=2e-2, rtol=1e-2)


IGEMM2_CASES = [
    # n, cin, cout, h, w, k, stride, pad, dil
    (2, 64, 128, 33, 33, 1, 1, 0, 1),
    (3, 128, 320, 33, 31, 3, 1, 1, 1),      # Cout tail for every tile width
    (1, 256, 256, 33, 33, 3, 1, 6, 6),
    (3, 512, 64, 9, 9, 3, 1, 18, 18),       # halo larger than the map
    (2, 64, 128, 65, 65, 3, 2, 1, 1),       # strided: dgrad runs the sdiv=2 tap-parity path
    (2, 128, 256, 33, 35, 1, 2, 0, 1),      # 1x1 stride 2 (projection shortcut)
    (2, 304, 256, 17, 17, 3, 1, 1, 1),      # Cin tail (not a multiple of 64)
    (2, 72, 48, 19, 19, 1, 1, 0, 1),        # Cin tail on the 1x1 fast path, Cout = 48
    (5, 64, 64, 129, 129, 3, 1, 1, 1),      # many M tiles, M tail
]


@pytest.mark.parametrize("cfg", [1, 3, 4, 5, 7, 8, 9, 10, 13, 14, 15, 16])
@pytest.mark.parametrize("case", IGEMM2_CASES)
def test_igemm2_all_tile_configs(case, cfg):
    """csrc/conv_igemm2.hip: every tile configuration, forward + data-gradient + fused statistics,
    against an fp32 CPU convolution."""
    from distributedpytorch_amd.ops import _hip_conv
    n, cin, cout, h, w, k, stride, pad, dil = case
    torch.manual_seed(3)
    x = _bf(torch.randn(n, cin, h, w, device=DEV))
    wt = _bf(torch.randn(cout, cin, k, k, device=DEV) / math.sqrt(cin * k * k))
    xr = x.float().cpu().requires_grad_(True)
    yr = F.conv2d(xr, wt.float().cpu(), None, stride, pad, dil)
    ho, wo = yr.shape[2:]
    gy = _bf(torch.randn_like(yr)).to(DEV)
    yr.backward(gy.float().cpu())
    xh, wh = _nhwc(x), _nhwc(wt)
    reps = 3
    stats = torch.zeros(reps, 2, cout, device=DEV)
    y = _hip_conv.igemm(xh, wh, (ho, wo), stride, 1, (pad, pad), (dil, dil), None, stats, cfg=cfg)
    assert _rel(y, yr) < 1.5e-2
    s = stats.sum(0).cpu()
    yd = yr.detach()
    # Σy cancels heavily: bound the error by the mass it was summed from
    assert float((s[0] - yd.sum(dim=(0, 2, 3))).abs().max()) < 5e-3 * float(yd.abs().sum(dim=(0, 2, 3)).max())
    assert _rel(s[1], (yd * yd).sum(dim=(0, 2, 3))) < 5e-3
    bias = torch.randn(cout, device=DEV)
    yb = _hip_conv.igemm(xh, wh, (ho, wo), stride, 1, (pad, pad), (dil, dil), bias, None, cfg=cfg)
    assert _rel(yb, yr.detach() + bias.cpu().view(1, -1, 1, 1)) < 1.5e-2
    wT = _hip_conv.weight_transpose(wh)
    pad_t = dil * (k - 1) - pad
    dx = _hip_conv.igemm(_nhwc(gy), wT, (h, w), 1, stride, (pad_t, pad_t), (dil, dil), cfg=cfg)
    assert _rel(dx, xr.grad) < 1.5e-2


WGRAD_CASES = [
    # n, cin, cout, h, w, k, stride, pad, dil
    (3, 64, 128, 33, 31, 1, 1, 0, 1),       # 1x1 fast path, pixel tail (3069 = 47*64 + 61)
    (2, 72, 200, 19, 19, 1, 1, 0, 1),       # Cout / Cin tails of the 128x128 tile
    (2, 64, 136, 33, 33, 3, 1, 2, 2),       # taps straddle tiles (NF = 576), dilation
    (2, 128, 64, 35, 33, 3, 2, 1, 1),       # strided
    (3, 256, 64, 9, 9, 3, 1, 12, 12),       # halo larger than the map
    (2, 64, 64, 21, 23, 1, 2, 0, 1),        # 1x1 stride 2 (general path with one tap)
    # K-steps of zeros skipped (two wave groups, Cin % 128 == 0: a block's weight columns lie in one tap)
    (3, 128, 256, 33, 33, 3, 1, 18, 18),    # ASPP rate 18 on 33x33: 18 of 33 rows are padding for the outer tap rows
    (2, 256, 256, 65, 65, 3, 1, 36, 36),    # Xception-65 OS8 ASPP rate 36 on 65x65
    (3, 128, 256, 33, 31, 3, 1, 12, 12),    # rate 12, odd width
    (2, 128, 256, 17, 17, 3, 2, 6, 6),      # strided and dilated
    (5, 128, 256, 5, 7, 3, 1, 2, 2),        # tiny map: every step runs over several images (never skipped)
    (2, 128, 384, 33, 33, 3, 1, 40, 40),    # halo larger than the map: only the centre tap sees pixels
]


@pytest.mark.parametrize("splits", [0, 1, 3, 5])
@pytest.mark.parametrize("case", WGRAD_CASES)
def test_wgrad_splits(case, splits):
    """csrc/conv_wgrad.hip: forced split-K counts (tail splits, one split) against an fp32 CPU
    weight gradient; also accumulation into a pre-filled gradient buffer."""
    from distributedpytorch_amd.ops import _hip_conv
    n, cin, cout, h, w, k, stride, pad, dil = case
    torch.manual_seed(5)
    x = _bf(torch.randn(n, cin, h, w, device=DEV))
    wr = torch.zeros(cout, cin, k, k, requires_grad=True)
    yr = F.conv2d(x.float().cpu(), wr, None, stride, pad, dil)
    gy = _bf(torch.randn_like(yr))
    yr.backward(gy.float())
    out = torch.ones(cout, cin, k, k, device=DEV).contiguous(memory_format=torch.channels_last)
    dw = _hip_conv.wgrad(_nhwc(gy.to(DEV)), _nhwc(x), (cout, cin, k, k), (stride, stride), (pad, pad), (dil, dil),
                         out=out, splits=splits)
    assert _rel(dw - 1.0, wr.grad) < 1e-2


@pytest.fixture
def dw_window(request):
    """Rolling-window depthwise kernels on / off for one test (csrc/depthwise.hip, dp_dwconv_set_window); restored."""
    from distributedpytorch_amd.ops import _hip_conv
    lib = _hip_conv._lib()
    old = lib.dp_dwconv_set_window(request.param)
    yield request.param
    lib.dp_dwconv_set_window(old)


@pytest.mark.parametrize("dw_window", [1, 0], indirect=True)
@pytest.mark.parametrize("stride,dil", [(1, 1), (1, 2), (2, 1), (1, 4)])
@pytest.mark.parametrize("shape", [(2, 72, 33, 31), (3, 136, 65, 65), (1, 64, 7, 129), (1, 64, 9, 200),
                                   (5, 64, 3, 5)])
def test_depthwise(stride, dil, shape, dw_window):
    """The rolling-window kernels (stride 1: several bands per image, column bands for maps wider than 160 columns,
    partial 64-channel slab, one to four rows per step) and the general-shape kernels behind them (stride 2; every
    shape with the window kernels switched off) against an fp32 CPU grouped convolution."""
    torch.manual_seed(1)
    n, c, h, w = shape
    x = _bf(torch.randn(n, c, h, w, device=DEV))
    wt = _bf(torch.randn(c, 1, 3, 3, device=DEV) / 3)
    xr, wr = x.float().cpu().requires_grad_(True), wt.float().cpu().requires_grad_(True)
    yr = F.conv2d(xr, wr, None, stride, dil, dil, groups=c)
    gy = _bf(torch.randn_like(yr)).to(DEV)
    yr.backward(gy.float().cpu())
    xh, wh = _nhwc(x).requires_grad_(True), wt.float().requires_grad_(True)
    yh, stats = ops.conv2d(xh, wh, None, stride, dil, dil, groups=c, with_stats=True)
    assert _rel(yh, yr) < 1.5e-2
    st = stats.sum(0).cpu()                  # fused in the tiled kernel, a separate pass behind the flat ones
    assert _rel(st[0], yr.detach().sum(dim=(0, 2, 3))) < 2e-2
    assert _rel(st[1], (yr.detach() ** 2).sum(dim=(0, 2, 3))) < 2e-2
    yh.backward(_nhwc(gy))
    assert _rel(xh.grad, xr.grad) < 1.5e-2
    assert _rel(wh.grad, wr.grad) < 1.5e-2


def test_conv_fused_statistics():
    """Σy, Σy² from the conv epilogue (replicated accumulators) == statistics of the output."""
    torch.manual_seed(20)
    for (n, cin, cout, hw, k) in [(4, 64, 128, 65, 3), (2, 256, 64, 33, 1), (16, 64, 256, 65, 1)]:
        x = _nhwc(_bf(torch.randn(n, cin, hw, hw, device=DEV)))
        w = _bf(torch.randn(cout, cin, k, k, device=DEV) / math.sqrt(cin * k * k)).float()
        y, stats = ops.conv2d(x, w, None, 1, k // 2, 1, with_stats=True)
        s = stats.sum(0)
        yf = F.conv2d(x.float().cpu(), w.cpu(), None, 1, k // 2, 1)
        assert _rel(s[0], yf.sum(dim=(0, 2, 3))) < 2e-3
        assert _rel(s[1], (yf * yf).sum(dim=(0, 2, 3))) < 2e-3


def test_conv_bn_act_fused_matches_unfused():
    torch.manual_seed(21)
    conv = ops.Conv2d(64, 128, 3, padding=1, bias=False).to(DEV)
    bn = ops.BatchNorm2d(128).to(DEV)
    x = _nhwc(_bf(torch.randn(4, 64, 33, 33, device=DEV)))
    res = _nhwc(_bf(torch.randn(4, 128, 33, 33, device=DEV)))
    outs = []
    for fused in (True, False):
        conv.zero_grad(); bn.zero_grad()
        bn.running_mean.zero_(); bn.running_var.fill_(1)
        xi = x.clone().requires_grad_(True)
        if fused:
            y = ops.conv_bn_act(conv, bn, xi, relu=True, residual=res)
        else:
            y = bn(conv(xi), relu=True, residual=res)
        y.float().square().mean().backward()
        outs.append((y.detach(), xi.grad, conv.weight.grad.clone(), bn.weight.grad.clone(), bn.running_var.clone()))
    for a, b in zip(*outs):
        assert _rel(a, b) < 2e-2


# ----------------------------------------------------------------------------- batch norm
@pytest.mark.parametrize("shape", [(2, 64, 33, 33), (3, 48, 17, 9), (2, 2048, 9, 9), (2, 728, 17, 17),
                                   (1, 4096, 5, 5)])
@pytest.mark.parametrize("relu,res", [(False, False), (True, False), (True, True), (False, True)])
def test_batchnorm_train(shape, relu, res):
    torch.manual_seed(2)
    n, c, h, w = shape
    x = _bf(torch.randn(shape, device=DEV) * 2 + 0.5)
    r = _bf(torch.randn(shape, device=DEV)) if res else None
    wt = torch.rand(c, device=DEV) + 0.5
    bs = torch.randn(c, device=DEV) * 0.1
    gy = _bf(torch.randn(shape, device=DEV))

    def run(dev, xin, rin):
        rm, rv = torch.zeros(c, device=dev), torch.ones(c, device=dev)
        xi = xin.to(dev).clone().requires_grad_(True)
        ri = rin.to(dev).clone().requires_grad_(True) if rin is not None else None
        wi, bi = wt.to(dev).clone().requires_grad_(True), bs.to(dev).clone().requires_grad_(True)
        y = ops.batch_norm_act(xi, wi, bi, rm, rv, True, 0.1, 1e-5, relu, ri)
        y.backward(gy.to(dev).to(y.dtype))
        return y, xi.grad, wi.grad, bi.grad, (ri.grad if ri is not None else None), rm, rv

    # reference: the plain tensor-math path of the same Function, fp32 on the CPU
    ref = run("cpu", x.float(), r.float() if res else None)
    got = run(DEV, _nhwc(x), _nhwc(r) if res else None)
    assert _rel(got[0], ref[0]) < 2e-2, "y"
    assert _rel(got[1], ref[1]) < 3e-2, "dx"
    assert _rel(got[2], ref[2]) < 2e-2, "dweight"
    assert _rel(got[3], ref[3]) < 2e-2, "dbias"
    if res:
        assert _rel(got[4], ref[4]) < 2e-2
    assert _rel(got[5], ref[5]) < 1e-2 and _rel(got[6], ref[6]) < 1e-2


@pytest.mark.parametrize("p", [0.5, 0.1])
@pytest.mark.parametrize("shape", [(2, 64, 33, 31), (3, 256, 17, 17)])
def test_batchnorm_fused_dropout(shape, p):
    """Dropout inside the BN+ReLU kernels (csrc/bn_relu.hip, struct Drop): the kept fraction, the
    1/(1-p) scale, and a backward that recomputes the SAME mask from the hash — checked against
    torch autograd run with the mask read back from the fused output."""
    torch.manual_seed(5)
    n, c, h, w = shape
    x = _nhwc(_bf(torch.randn(shape, device=DEV) * 2 + 0.3))
    bn = ops.BatchNorm2d(c).to(DEV).train()
    with torch.no_grad():
        bn.weight.uniform_(0.5, 1.5)
        bn.bias.uniform_(-0.5, 0.5)
    xh = x.clone().requires_grad_(True)
    y = bn(xh, relu=True, dropout=p)
    gy = _nhwc(_bf(torch.randn_like(y)))
    y.backward(gy)

    xr = x.float().requires_grad_(True)
    wr = bn.weight.detach().clone().requires_grad_(True)
    br = bn.bias.detach().clone().requires_grad_(True)
    act = F.relu(F.batch_norm(xr, None, None, wr, br, True, 0.1, bn.eps))
    live = act.detach() > 1e-3                         # away from the ReLU edge (bf16 rounding of y)
    keep = (y.detach().float() != 0)
    frac = float(keep[live].float().mean())
    assert abs(frac - (1 - p)) < 0.02, frac
    p_eff = round(p * 65536) / 65536
    ref = act * keep / (1 - p_eff)
    assert _rel(y, ref) < 1.5e-2
    ref.backward(gy.float())
    assert _rel(xh.grad, xr.grad) < 2e-2
    assert _rel(bn.weight.grad, wr.grad) < 1e-2
    assert _rel(bn.bias.grad, br.grad) < 1e-2
    # a second call draws a different mask
    y2 = bn(x, relu=True, dropout=p)
    assert float(((y2 != 0) != (y.detach() != 0)).float().mean()) > 0.05


def test_batchnorm_eval_and_frozen():
    torch.manual_seed(3)
    c = 64
    x = _bf(torch.randn(2, c, 17, 17, device=DEV))
    bn = ops.BatchNorm2d(c).to(DEV)
    bn.running_mean.normal_()
    bn.running_var.uniform_(0.5, 2)
    bn.eval()
    y = bn(_nhwc(x), relu=True)
    ref = F.relu(F.batch_norm(x.float(), bn.running_mean, bn.running_var, bn.weight, bn.bias, False))
    assert _rel(y, ref) < 2e-2
    bn.train()
    bn.frozen = True
    xh = _nhwc(x).requires_grad_(True)
    bn(xh, relu=True).float().sum().backward()
    xr = x.float().requires_grad_(True)
    F.relu(F.batch_norm(xr, bn.running_mean, bn.running_var, bn.weight, bn.bias, False)).sum().backward()
    assert _rel(xh.grad, xr.grad) < 2e-2


# ----------------------------------------------------------------------------- loss / resize / pool
@pytest.mark.parametrize("c,hw", [(21, 65), (21, 513), (7, 33)])
def test_softmax_ce(c, hw):
    torch.manual_seed(4)
    n = 2
    logits = _bf(torch.randn(n, c, hw, hw, device=DEV) * 3)
    tgt = torch.randint(0, c, (n, hw, hw), device=DEV)
    tgt[0, : hw // 4] = 255
    lr = logits.float().requires_grad_(True)
    ref = F.cross_entropy(lr, tgt, ignore_index=255)
    (ref * 0.5).backward()
    lh = _nhwc(logits).requires_grad_(True)
    got = ops.softmax_cross_entropy(lh, tgt.to(torch.uint8), 255)
    (got * 0.5).backward()
    assert abs(float(got) - float(ref)) < 2e-3 * max(1.0, abs(float(ref)))
    assert _rel(lh.grad, lr.grad) < 2e-2
    # all-ignored batch: loss 0, gradient 0, no NaN
    tgt.fill_(255)
    lh2 = _nhwc(logits).requires_grad_(True)
    z = ops.softmax_cross_entropy(lh2, tgt, 255)
    z.backward()
    assert float(z) == 0.0 and float(lh2.grad.abs().max()) == 0.0


@pytest.mark.parametrize("c,hin,hout,padded", [(21, (129, 129), (513, 513), True), (21, (33, 17), (129, 65), False),
                                               (7, (17, 17), (33, 33), False), (21, (33, 33), (100, 77), True),
                                               (21, (17, 17), (129, 129), False), (3, (1, 1), (4, 4), False)])
def test_upsample_ce_fused(c, hin, hout, padded):
    """csrc/upsample_ce.hip: CE of the bilinearly upsampled logits, loss and gradient, against
    F.interpolate(align_corners=True) + F.cross_entropy in fp32.  (17 -> 129 is outside the fused
    kernels' footprint budget and runs the unfused pair through the same entry point.)"""
    torch.manual_seed(11)
    n = 2
    base = _bf(torch.randn(n, c, *hin, device=DEV) * 3)
    t = torch.randint(0, c, (n, *hout), device=DEV)
    t[:, : max(hout[0] // 7, 1)] = 255
    t[0, -1, -1] = 255
    xr = base.float().cpu().requires_grad_(True)
    up = F.interpolate(xr, size=hout, mode="bilinear", align_corners=True)
    lr = F.cross_entropy(up, t.cpu(), ignore_index=255)
    (lr * 1.7).backward()
    if padded:   # the classifier's output: a [:, :c] view of a channel-padded NHWC buffer
        cp = (c + 7) // 8 * 8
        buf = torch.full((n, cp, *hin), 9.0, device=DEV, dtype=torch.bfloat16).contiguous(memory_format=torch.channels_last)
        buf[:, :c] = base
        xh = buf[:, :c].detach().requires_grad_(True)
    else:
        xh = _nhwc(base).requires_grad_(True)
    for tt in (t, t.to(torch.uint8)):
        xh.grad = None
        lh = ops.softmax_cross_entropy(xh, tt, 255)
        (lh * 1.7).backward()
        assert abs(float(lh) - float(lr)) < 2e-3 * max(1.0, abs(float(lr)))
        assert _rel(xh.grad, xr.grad) < 1.5e-2
    # all-ignored batch: loss 0, gradient 0
    xh.grad = None
    lz = ops.softmax_cross_entropy(xh, torch.full_like(t, 255), 255)
    lz.backward()
    assert float(lz) == 0.0 and float(xh.grad.abs().max()) == 0.0


@pytest.mark.parametrize("c,hin,hout", [(256, 33, 129), (21, 129, 513), (48, 17, 65), (21, 33, 20), (64, 1, 9), (1, 65, 513),
                                          (3, 9, 40), (5, 20, 301)])
def test_bilinear(c, hin, hout):
    torch.manual_seed(5)
    x = _bf(torch.randn(2, c, hin, hin, device=DEV))
    xr = x.float().requires_grad_(True)
    yr = F.interpolate(xr, size=(hout, hout), mode="bilinear", align_corners=True)
    gy = _bf(torch.randn_like(yr))
    yr.backward(gy.float())
    xh = _nhwc(x).requires_grad_(True)
    yh = ops.bilinear_resize(xh, (hout, hout))
    assert _rel(yh, yr) < 1e-2
    yh.backward(_nhwc(gy))
    assert _rel(xh.grad, xr.grad) < 1.5e-2


def test_pools():
    torch.manual_seed(6)
    x = _bf(torch.randn(2, 64, 65, 67, device=DEV))
    xr = x.float().requires_grad_(True)
    yr = F.max_pool2d(xr, 3, 2, 1)
    gy = _bf(torch.randn_like(yr))
    yr.backward(gy.float())
    xh = _nhwc(x).requires_grad_(True)
    yh = ops.max_pool3x3s2(xh)
    assert torch.equal(yh.float(), yr)
    yh.backward(_nhwc(gy))
    assert _rel(xh.grad, xr.grad) < 1e-2
    x2 = _bf(torch.randn(3, 2048, 9, 9, device=DEV))
    x2r = x2.float().requires_grad_(True)
    pr = F.adaptive_avg_pool2d(x2r, 1)
    g2 = _bf(torch.randn_like(pr))
    pr.backward(g2.float())
    x2h = _nhwc(x2).requires_grad_(True)
    ph = ops.global_avg_pool(x2h)
    assert _rel(ph, pr) < 1e-2
    ph.backward(g2)
    assert _rel(x2h.grad, x2r.grad) < 1e-2


def test_sgd_flat_matches_torch():
    torch.manual_seed(7)
    nel = 100003
    p0 = torch.randn(nel, device=DEV)
    ref_p = p0.clone().requires_grad_(True)
    opt = torch.optim.SGD([ref_p], lr=0.1, momentum=0.9, weight_decay=5e-4)
    p, g, m = p0.clone(), torch.zeros(nel, device=DEV), torch.zeros(nel, device=DEV)
    p16 = torch.zeros(nel, device=DEV, dtype=torch.bfloat16)
    for step in range(3):
        grad = torch.randn(nel, device=DEV)
        ref_p.grad = grad.clone() * 0.5
        opt.step()
        g.copy_(grad)
        ops.sgd_flat_step_(p, g, m, p16, 0.1, 0.9, 5e-4, grad_scale=0.5, zero_grad=True, first_step=(step == 0))
        assert float(g.abs().max()) == 0.0
    assert torch.allclose(p, ref_p.detach(), atol=1e-5, rtol=1e-5)
    assert torch.equal(p16, p.to(torch.bfloat16))


def test_affine_augment_matches_reference():
    torch.manual_seed(8)
    bsz, h, w = 3, 65, 77
    img = torch.randint(0, 256, (bsz, h, w, 3), dtype=torch.uint8, device=DEV)
    lab = torch.randint(0, 21, (bsz, h, w), dtype=torch.uint8, device=DEV)
    mats = ops.make_affine_params(bsz, (h, w), (49, 49), scales=(0.5, 2.0), rots=(-10, 10),
                                  rng=np.random.default_rng(0))
    out, ol = ops.affine_augment(img, lab, mats, (49, 49))
    with ops.backend("torch"):
        ref, rl = ops.affine_augment(img, lab, mats, (49, 49), out_dtype=torch.float32)
    assert out.shape == (bsz, 3, 49, 49) and out.is_contiguous(memory_format=torch.channels_last)
    assert float((out.float() - ref).abs().max()) < 0.06
    assert float((ol != rl).float().mean()) < 0.01   # nearest ties at .5 may differ


def test_confusion_matrix():
    torch.manual_seed(9)
    k = 21
    logits = _bf(torch.randn(2, k, 65, 65, device=DEV))
    tgt = torch.randint(0, k, (2, 65, 65), device=DEV)
    tgt[:, :3] = 255
    cm = ops.confusion_matrix(_nhwc(logits), tgt, k)
    with ops.backend("torch"):
        ref = ops.confusion_matrix(logits.float(), tgt, k)
    assert torch.equal(cm, ref)
    pred = logits.float().argmax(1).to(torch.uint8)
    cm2 = ops.confusion_matrix(pred, tgt.to(torch.uint8), k)
    assert torch.equal(cm2, ref)


# ----------------------------------------------------------------------------- end to end
def test_model_step_matches_stock_path():
    """One DeepLabv3+ (R18) training step through the HIP kernels vs the stock-op path."""
    from distributedpytorch_amd.models import build_model
    torch.manual_seed(10)
    x = torch.randn(2, 3, 129, 129, device=DEV)
    y = torch.randint(0, 21, (2, 129, 129), device=DEV)
    y[:, :5] = 255
    from distributedpytorch_amd.ops import batchnorm as bn_mod
    losses = {}
    for backend in ("torch", "auto"):
        torch.manual_seed(11)
        m = build_model("deeplabv3plus_resnet18", 21).to(DEV).to(memory_format=torch.channels_last)
        m.train()  # batch statistics keep activations O(1); same seed -> same dropout masks
        torch.manual_seed(12)
        # stock side: ATen kernels only (MIOpen's bf16 NHWC train-mode batch-norm segfaults on this
        # small shape on ROCm 7.0/7.2 — host-side crash inside F.batch_norm, observed on MI355X)
        prev = torch.backends.cudnn.enabled
        torch.backends.cudnn.enabled = False
        prev_fuse, bn_mod._FUSE_DROPOUT = bn_mod._FUSE_DROPOUT, False     # torch's RNG on both sides: same masks
        try:
            with ops.backend(backend):
                out = m(_nhwc(_bf(x)))
                loss = ops.softmax_cross_entropy(out, y)
        finally:
            torch.backends.cudnn.enabled = prev
            bn_mod._FUSE_DROPOUT = prev_fuse
        losses[backend] = float(loss.detach())
    assert abs(losses["auto"] - losses["torch"]) < 0.05 * abs(losses["torch"])


def test_engine_trains():
    from distributedpytorch_amd.models import build_model
    from distributedpytorch_amd.parallel import TrainEngine
    torch.manual_seed(12)
    m = build_model("deeplabv3plus_resnet18", 21)
    eng = TrainEngine(m, ops.SoftmaxCrossEntropy(255), lr=0.01, device=torch.device(DEV))
    x = torch.randn(4, 3, 65, 65, device=DEV)
    y = torch.randint(0, 21, (4, 65, 65), device=DEV)
    first = float(eng.train_step(x, y))
    for _ in range(15):
        last = float(eng.train_step(x, y))
    assert math.isfinite(last) and last < first


def test_batched_weight_transpose_matches_per_layer():
    """parallel.arena keeps tap-flipped, Cin<->Cout transposed bf16 weights for the data-gradient
    (csrc/conv_igemm.hip weight_transpose_batched_kernel): equal to the per-layer transpose, before
    and after an optimiser step."""
    from distributedpytorch_amd.models import build_model
    from distributedpytorch_amd.ops import SoftmaxCrossEntropy, _hip_conv
    from distributedpytorch_amd.parallel import TrainEngine
    torch.manual_seed(0)
    model = build_model("deeplabv3plus_resnet18", 21)
    eng = TrainEngine(model, SoftmaxCrossEntropy(255), lr=0.01, device=torch.device(DEV))

    def check():
        n = 0
        for p in model.parameters():
            wt = getattr(p, "_dpamd_lowpT", None)
            if wt is None:
                continue
            assert p._dpamd_lowpT_ok
            ref = _hip_conv.weight_transpose(p._dpamd_lowp)
            assert wt.shape == ref.shape and torch.equal(wt.contiguous(memory_format=torch.channels_last), ref)
            n += 1
        assert n > 20
    check()
    x = torch.randn(2, 3, 65, 65, device=DEV)
    y = torch.randint(0, 21, (2, 65, 65), device=DEV)
    eng.train_step(x, y)
    check()


def test_graphed_eval_forward_matches_eager():
    """parallel.graphs.GraphedForward: the hipGraph replay of the batch-1 inference forward equals
    the eager forward, also after the weights changed in place (graphs hold pointers)."""
    from distributedpytorch_amd.models import build_model
    from distributedpytorch_amd.parallel import GraphedForward
    torch.manual_seed(0)
    model = build_model("deeplabv3plus_resnet18", 21).to(DEV).to(memory_format=torch.channels_last).eval()
    g = GraphedForward(model)
    for trial in range(3):
        x = _nhwc(_bf(torch.randn(1, 3, 129, 129, device=DEV)))
        with torch.no_grad():
            ref = model(x).float()
        got = g(x).float()
        assert torch.equal(got, ref)
        with torch.no_grad():
            for p in model.parameters():
                p.mul_(0.97)


def test_bn_dropout_backward_prologue_in_dgrad_matches_separate_kernels():
    """The decoder's conv-BN-ReLU-Dropout stages (models/deeplab.py, _LastConv): the consumer's data-gradient kernel
    applies mask (from the dropped output itself: a > 0 = positive and kept), dropout scale and the two BN-backward
    reductions (Conv2Args::mask_scale) — same gradients as the separate bn_bwd_reduce pass that recomputes the keep
    decisions from the seed.  Also the 21-class classifier: its data gradient runs with the class channels padded
    to 24, and still carries the prologue of the BN in front of it."""
    from distributedpytorch_amd.models.deeplab import _LastConv
    from distributedpytorch_amd.ops import _hip_conv, batchnorm as bn_mod

    def grads(flag):
        old = _hip_conv._BN_PROLOGUE
        _hip_conv._BN_PROLOGUE = flag
        try:
            torch.manual_seed(0)
            bn_mod._drop_calls = 100                     # the same masks on both paths
            net = _LastConv(64, 21, mid=64).to(DEV).to(memory_format=torch.channels_last).train()
            with torch.no_grad():
                for m in net.modules():
                    if hasattr(m, "running_mean"):
                        m.weight.uniform_(0.6, 1.4); m.bias.normal_(0, 0.5)
            x = _nhwc(_bf(torch.randn(4, 64, 33, 31, device=DEV))).requires_grad_(True)
            t = _nhwc(_bf(torch.randn(4, 21, 33, 31, device=DEV)))
            (net(x, True).float() - t.float()).pow(2).mean().backward()
            g = {n: p.grad.detach().float().clone() for n, p in net.named_parameters()}
            g["input"] = x.grad.detach().float().clone()
            return g
        finally:
            _hip_conv._BN_PROLOGUE = old
    counts = []
    orig = _hip_conv.dgrad_bn_fused
    _hip_conv.dgrad_bn_fused = lambda *a, **k: (counts.append(a[7].get("mask_scale")), orig(*a, **k))[1]
    try:
        g1 = grads(True)
    finally:
        _hip_conv.dgrad_bn_fused = orig
    assert len(counts) == 2 and abs(counts[0] - 1 / 0.9) < 1e-3 and abs(counts[1] - 2.0) < 1e-3, counts
    g0, g00 = grads(False), grads(False)
    worst = max(_rel(g1[k], g0[k]) for k in g0)
    noise = max(_rel(g00[k], g0[k]) for k in g0)
    assert worst < max(3e-2, 8 * noise), (worst, noise, {k: _rel(g1[k], g0[k]) for k in g0})


@pytest.mark.parametrize("block_name,hw", [("Bottleneck", (33, 31)), ("BasicBlock", (33, 31)), ("Bottleneck", (65, 67))])
def test_bn_backward_prologue_in_dgrad_matches_separate_kernels(block_name, hw):
    """csrc/conv_igemm2.hip (Conv2Args::bnx): BN+ReLU backward's mask and reductions computed in the
    consumer's data-gradient store loop give the same parameter and input gradients as the separate
    bn_bwd_reduce pass.  One residual stage (projection block + 2 identity blocks) keeps the
    comparison well conditioned (a whole random-init ResNet at batch 2 is chaotic: two runs of the
    SAME path differ by O(1) once the autotuner picks different tiles).  The 65x67 case has enough rows
    (17420) for the store loop to spread its sums over two replicas, which bn_bwd_apply folds."""
    from distributedpytorch_amd.models import resnet
    from distributedpytorch_amd.ops import _hip_conv
    block = getattr(resnet, block_name)

    def grads(flag):
        old = _hip_conv._BN_PROLOGUE
        _hip_conv._BN_PROLOGUE = flag
        try:
            torch.manual_seed(0)
            cin, planes = 64, 32
            cout = planes * block.expansion
            blocks = [block(cin, planes, 1, 1, resnet._Projection(cin, cout, 1)), block(cout, planes), block(cout, planes)]
            blocks[1].input_exclusive = blocks[2].input_exclusive = True
            net = torch.nn.Sequential(*blocks).to(DEV).to(memory_format=torch.channels_last).train()
            x = _nhwc(_bf(torch.randn(4, cin, *hw, device=DEV))).requires_grad_(True)
            t = _nhwc(_bf(torch.randn(4, cout, *hw, device=DEV)))
            (net(x).float() - t.float()).pow(2).mean().backward()
            g = {n: p.grad.detach().float().clone() for n, p in net.named_parameters()}
            g["input"] = x.grad.detach().float().clone()
            return g
        finally:
            _hip_conv._BN_PROLOGUE = old
    g1, g0, g00 = grads(True), grads(False), grads(False)
    worst = max(_rel(g1[k], g0[k]) for k in g0)
    noise = max(_rel(g00[k], g0[k]) for k in g0)     # atomics order -> bf16 rounding flips, amplified by BN
    # a wrong mask or a missing term shows up as O(1); rounding-order differences stay at a few percent
    assert worst < max(6e-2, 8 * noise), (worst, noise)


@pytest.mark.parametrize("cfg", [1, 5, 9, 14, 15, 16])
@pytest.mark.parametrize("k,with_res", [(1, True), (3, False)])
def test_dgrad_bn_fused_store_loop_per_tile_config(cfg, k, with_res):
    """csrc/conv_igemm2.hip, Conv2Args::bnx, for a forced tile configuration (the small-LDS tiles 14-18
    included): masked gradient dm = (dgrad [+ skip])·[a > 0] and the partial sums Σdm, Σdm·x̂ against a
    plain fp32 computation from an fp32 CPU transposed convolution."""
    from distributedpytorch_amd.ops import _hip_conv
    torch.manual_seed(5)
    n, cin, cout, h, w = 3, 96, 160, 37, 29          # forward conv cin -> cout; its dgrad produces cin channels
    pad = k // 2
    wt = _bf(torch.randn(cout, cin, k, k, device=DEV) / math.sqrt(cin * k * k))
    dy = _bf(torch.randn(n, cout, h, w, device=DEV))
    bx = _bf(torch.randn(n, cin, h, w, device=DEV))                      # pre-BN activations of the producer
    mean, invstd = torch.randn(cin, device=DEV) * 0.1, torch.rand(cin, device=DEV) + 0.5
    gam, bet = torch.rand(cin, device=DEV) + 0.5, torch.randn(cin, device=DEV) * 0.2
    res = _bf(torch.randn(n, cin, h, w, device=DEV)) if with_res else None
    v = lambda t: t.view(1, -1, 1, 1)  # noqa: E731
    pre = bx.float() * v(gam * invstd) + v(bet - mean * gam * invstd)
    a = _bf(torch.relu(pre + (res.float() if with_res else 0)))          # the activation this conv consumed
    da = torch.nn.grad.conv2d_input((n, cin, h, w), wt.float().cpu(), dy.float().cpu(), 1, pad, 1).to(DEV)
    if with_res:
        da = da + res.float()
    mask = (a.float() > 0) if with_res else (pre > 0)
    dm_ref = da * mask
    holder = dict(x=_nhwc(bx), mask_from_a=with_res, mean=mean, invstd=invstd, w=gam, b=bet)
    wT = _hip_conv.weight_transpose(_nhwc(wt))
    old = _hip_conv._FORCED_CFG
    _hip_conv._FORCED_CFG = cfg
    try:
        out = _hip_conv.dgrad_bn_fused(_nhwc(dy), wT, (h, w), 1, (pad, pad), (1, 1),
                                       _nhwc(res) if with_res else None, holder, _nhwc(a) if with_res else None)
    finally:
        _hip_conv._FORCED_CFG = old
    assert out is not None
    dm, part = out
    assert part.dim() == 3 and tuple(part.shape[1:]) == (2, cin)
    assert _rel(dm, dm_ref) < 1.5e-2
    dmf = dm.float()                                   # the sums are taken over the bf16-rounded dm
    xhat = (bx.float() - v(mean)) * v(invstd)
    sums = part.sum(0)
    s1_ref, s2_ref = dmf.sum(dim=(0, 2, 3)), (dmf * xhat).sum(dim=(0, 2, 3))
    scale = float(dmf.abs().sum(dim=(0, 2, 3)).max())
    assert float((sums[0] - s1_ref).abs().max()) < 2e-3 * scale
    assert float((sums[1] - s2_ref).abs().max()) < 4e-3 * scale


@pytest.mark.parametrize("cfg", [5, 9, 16])
def test_dgrad_bn_fused_mask_bits_equal_activation_mask(cfg):
    """The positivity bit mask a residual block's BatchNorm apply kernel leaves behind (csrc/bn_relu.hip,
    ApplyArgs::mask; 1 bit per element) gives the fused data-gradient store loop (csrc/conv_igemm2.hip,
    Conv2Args::bnmask) exactly the mask it used to take from re-reading the activation: same dm, same sums;
    and the bits themselves are [a > 0] of the kernel's own output."""
    from distributedpytorch_amd.ops import _hip_bn, _hip_conv
    torch.manual_seed(11)
    n, cin, cout, h, w = 3, 96, 160, 37, 29
    wt = _bf(torch.randn(cout, cin, 1, 1, device=DEV) / math.sqrt(cin))
    dy = _nhwc(_bf(torch.randn(n, cout, h, w, device=DEV)))
    bx = _nhwc(_bf(torch.randn(n, cin, h, w, device=DEV)))
    res = _nhwc(_bf(torch.randn(n, cin, h, w, device=DEV)))
    skip = _nhwc(_bf(torch.randn(n, cin, h, w, device=DEV)))
    gam, bet = torch.rand(cin, device=DEV) + 0.5, torch.randn(cin, device=DEV) * 0.2
    s, ss = _hip_bn.stats(bx)
    bits = torch.zeros(bx.numel() // 8, dtype=torch.uint8, device=DEV)
    a, mean, invstd = _hip_bn.apply_from_stats(bx, s.clone(), ss.clone(), gam, bet, res, None, None,
                                               float(n * h * w), 1e-5, 0.1, True, mask_out=bits)
    # bit j of byte (pixel, chunk) = element 8*chunk + j of that pixel is positive
    a_rows = a.permute(0, 2, 3, 1).reshape(-1, cin // 8, 8)
    want = ((a_rows.float() > 0).to(torch.int32) << torch.arange(8, device=DEV, dtype=torch.int32)).sum(-1)
    assert torch.equal(bits.view(-1, cin // 8).to(torch.int32), want)
    wT = _hip_conv.weight_transpose(_nhwc(wt))
    old = _hip_conv._FORCED_CFG
    _hip_conv._FORCED_CFG = cfg
    try:
        base = dict(x=bx, mask_from_a=True, mean=mean, invstd=invstd, w=gam, b=bet)
        dm0, p0 = _hip_conv.dgrad_bn_fused(dy, wT, (h, w), 1, (0, 0), (1, 1), skip, dict(base), a)
        dm0, p0 = dm0.clone(), p0.sum(0).clone()
        dm1, p1 = _hip_conv.dgrad_bn_fused(dy, wT, (h, w), 1, (0, 0), (1, 1), skip, dict(base, mask_bits=bits), None)
    finally:
        _hip_conv._FORCED_CFG = old
    assert torch.equal(dm0, dm1)
    assert _rel(p1.sum(0), p0) < 1e-5


@pytest.mark.parametrize("C", [64, 728])
def test_bn_bwd_apply_folds_sum_replicas(C):
    """csrc/bn_relu.hip bn_bwd_apply_kernel: the [R, 2, C] partial sums a data-gradient store loop leaves
    behind (large maps spread their atomics over R replicas) are folded inside the kernel — same dx and
    same affine-gradient sinks as with the replicas summed beforehand by a separate kernel."""
    from distributedpytorch_amd.ops import _hip_bn
    torch.manual_seed(3)
    n, h, w, R = 2, 19, 23, 5
    x = _nhwc(_bf(torch.randn(n, C, h, w, device=DEV)))
    dy = _nhwc(_bf(torch.randn(n, C, h, w, device=DEV)))
    mean, invstd = torch.randn(C, device=DEV) * 0.1, torch.rand(C, device=DEV) + 0.5
    wgt, bias = torch.rand(C, device=DEV) + 0.5, torch.randn(C, device=DEV)
    part = torch.randn(R, 2, C, device=DEV)
    red = part.sum(0)
    count = float(n * h * w)
    gw0, gb0 = torch.zeros(C, device=DEV), torch.zeros(C, device=DEV)
    gw1, gb1 = torch.zeros(C, device=DEV), torch.zeros(C, device=DEV)
    ref, _ = _hip_bn.bwd_apply(dy, x, None, mean, invstd, wgt, bias, red[0], red[1], count, 0, False,
                               gw0, gb0, red[0], red[1])
    got, _ = _hip_bn.bwd_apply(dy, x, None, mean, invstd, wgt, bias, part[0, 0], part[0, 1], count, 0, False,
                               gw1, gb1, part[0, 0], part[0, 1], sum_replicas=part)
    assert _rel(got.float(), ref.float()) < 1e-2
    assert torch.allclose(gw1, red[1], atol=1e-5) and torch.allclose(gb1, red[0], atol=1e-5)
    assert torch.allclose(gw1, gw0, atol=1e-5) and torch.allclose(gb1, gb0, atol=1e-5)
    # the replica state is per launch: the next call reads plain [C] sums again
    again, _ = _hip_bn.bwd_apply(dy, x, None, mean, invstd, wgt, bias, red[0], red[1], count, 0, False)
    assert torch.equal(again, ref)
    with pytest.raises(ValueError):
        _hip_bn.bwd_apply(dy, x, None, mean, invstd, wgt, bias, red[0], red[1], count, 0, False, sum_replicas=part)



@pytest.mark.parametrize("train", [True, False])
def test_inplace_concat_matches_torch_cat(train):
    """ops.concat: ASPP + decoder with every producer writing its channel slice of the concat buffer
    (BN output row stride, bilinear output stride, broadcast) and every backward consumer reading its
    slice of the concat gradient through the stride — against the same modules run with torch.cat."""
    from distributedpytorch_amd.models.deeplab import ASPP, Decoder
    from distributedpytorch_amd.ops import batchnorm as bn_mod
    from distributedpytorch_amd.ops import concat

    def run(flag):
        old = concat.ENABLED
        concat.ENABLED = flag
        try:
            torch.manual_seed(21)
            bn_mod._drop_calls = 0                       # same fused-dropout seeds in both runs
            aspp, dec = ASPP(64, 16).to(DEV), Decoder(32, 21).to(DEV)
            for m in (aspp, dec):
                m.to(memory_format=torch.channels_last).train(train)
            feat = _nhwc(_bf(torch.randn(2, 64, 9, 11, device=DEV))).requires_grad_(True)
            low = _nhwc(_bf(torch.randn(2, 32, 33, 41, device=DEV))).requires_grad_(True)
            out = dec(aspp(feat), low)
            out.float().pow(2).mean().backward()
            g = {f"aspp.{n}": p.grad.float().clone() for n, p in aspp.named_parameters() if p.grad is not None}
            g.update({f"dec.{n}": p.grad.float().clone() for n, p in dec.named_parameters() if p.grad is not None})
            g["feat"], g["low"] = feat.grad.float().clone(), low.grad.float().clone()
            return out.detach().float(), g
        finally:
            concat.ENABLED = old
    o1, g1 = run(True)
    o0, g0 = run(False)
    o00, g00 = run(False)
    assert _rel(o1, o0) < 2e-2
    worst = max(_rel(g1[k], g0[k]) for k in g0)
    noise = max(_rel(g00[k], g0[k]) for k in g0)        # atomics order of the fused statistics
    assert worst < max(3e-2, 6 * noise), (worst, noise)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("heads", [1, 3])
def test_balanced_bce_multi_matches_the_tensor_op_formulation(dtype, heads):
    """csrc/loss_upsample.hip bce_* kernels: SegmentationMultiLosses(1) over DANet's heads
    (`/root/reference/train_pascal.py:119`, `:199`) in three launches == utils.losses.class_balanced_cross_entropy_loss
    per head in fp32 tensor ops, value and gradients; two runs bit-identical (per-block partial sums, no atomics)."""
    from distributedpytorch_amd.utils.losses import SegmentationMultiLosses, class_balanced_cross_entropy_loss
    g = torch.Generator().manual_seed(heads)
    outs = [(torch.randn(5, 1, 67, 45, generator=g) * 3).to(dtype).to(DEV).requires_grad_(True) for _ in range(heads)]
    target = (torch.rand(5, 1, 67, 45, generator=g) > 0.7).float().to(DEV)
    weights = (1.0, 0.4, 0.25)
    crit = SegmentationMultiLosses(1, weights)
    loss = crit(tuple(outs) if heads > 1 else outs[0], target)
    (loss * 2.5).backward()
    refs = [o.detach().float().clone().requires_grad_(True) for o in outs]
    want = sum(w * class_balanced_cross_entropy_loss(r, target, size_average=False, batch_average=True)
               for w, r in zip(weights, refs))
    (want * 2.5).backward()
    assert abs(float(loss) - float(want)) < 2e-4 * abs(float(want)) + 1e-5, (float(loss), float(want))
    for o, r in zip(outs, refs):
        assert o.grad.dtype == dtype
        assert _rel(o.grad.float(), r.grad) < (1e-5 if dtype == torch.float32 else 6e-3)
    outs2 = [o.detach().clone().requires_grad_(True) for o in outs]
    loss2 = crit(tuple(outs2) if heads > 1 else outs2[0], target)
    (loss2 * 2.5).backward()
    assert torch.equal(loss, loss2) and all(torch.equal(a.grad, b.grad) for a, b in zip(outs, outs2))
