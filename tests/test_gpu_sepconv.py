"""Folded separable-convolution unit (ops.sepconv: BatchNorm folded into the pointwise weights) against an fp32
CPU reference of the plain composition ``BN2(pointwise(BN1(depthwise(x)))) (+residual)(+ReLU)`` and against the
unfolded GPU composition.  Run on an MI355X: ``pytest -m gpu``."""
import pytest
import torch
import torch.nn.functional as F

from distributedpytorch_amd.models.xception import _Unit
from distributedpytorch_amd.ops import sepconv

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _cos(a, b):
    a, b = a.detach().double().flatten().cpu(), b.detach().double().flatten().cpu()
    return float((a @ b) / (a.norm() * b.norm()).clamp_min(1e-30))


def _rel(a, b):
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-12))


def _chan_rel(a, b, dim):
    """worst per-channel error relative to that channel's own magnitude"""
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    dims = [d for d in range(a.dim()) if d != dim]
    err = (a - b).abs().amax(dim=dims)
    mag = b.abs().amax(dim=dims).clamp_min(1e-3 * float(b.abs().max()))
    return float((err / mag).max())


def _reference(unit, x, gy, residual, relu, dil, stride):
    """fp32 CPU composition; returns (y, dx, dres, grads dict, running stats dict)."""
    sep = unit.sep
    P = {k: v.detach().float().cpu().clone().requires_grad_(True) for k, v in
         dict(dw=sep.conv1.weight, g1=sep.bn.weight, b1=sep.bn.bias, pw=sep.pointwise.weight, g2=unit.bn.weight,
              b2=unit.bn.bias).items()}
    # the kernels consume bf16 copies of the convolution weights
    dw = P["dw"].to(torch.bfloat16).float().detach().requires_grad_(True)
    pw = P["pw"].to(torch.bfloat16).float().detach().requires_grad_(True)
    xr = x.float().cpu().requires_grad_(True)
    rr = residual.float().cpu().requires_grad_(True) if residual is not None else None
    rm1, rv1 = torch.zeros(dw.shape[0]), torch.ones(dw.shape[0])
    rm2, rv2 = torch.zeros(pw.shape[0]), torch.ones(pw.shape[0])
    z = F.conv2d(xr, dw, None, stride, dil, dil, groups=dw.shape[0])
    a = F.batch_norm(z, rm1, rv1, P["g1"], P["b1"], True, 0.1, 1e-5)
    y = F.conv2d(a, pw)
    y = F.batch_norm(y, rm2, rv2, P["g2"], P["b2"], True, 0.1, 1e-5)
    if rr is not None:
        y = y + rr
    if relu:
        y = torch.relu(y)
    y.backward(gy.float().cpu())
    grads = dict(dw=dw.grad, g1=P["g1"].grad, b1=P["b1"].grad, pw=pw.grad, g2=P["g2"].grad, b2=P["b2"].grad)
    return y, xr.grad, (rr.grad if rr is not None else None), grads, dict(rm1=rm1, rv1=rv1, rm2=rm2, rv2=rv2)


def _run_gpu(unit, x, gy, residual, relu, fold):
    old = sepconv._FOLD
    sepconv._FOLD = fold
    try:
        for p in unit.parameters():
            p.grad = None
        for bn in (unit.sep.bn, unit.bn):
            bn.running_mean.zero_(); bn.running_var.fill_(1.0)
        xh = x.clone().requires_grad_(True)
        rh = residual.clone().requires_grad_(True) if residual is not None else None
        y = unit(xh, residual=rh, relu=relu)
        y.backward(gy)
        torch.cuda.synchronize()
        sep = unit.sep
        grads = dict(dw=sep.conv1.weight.grad, g1=sep.bn.weight.grad, b1=sep.bn.bias.grad, pw=sep.pointwise.weight.grad,
                     g2=unit.bn.weight.grad, b2=unit.bn.bias.grad)
        stats = dict(rm1=sep.bn.running_mean.clone(), rv1=sep.bn.running_var.clone(),
                     rm2=unit.bn.running_mean.clone(), rv2=unit.bn.running_var.clone())
        return y, xh.grad, (rh.grad if rh is not None else None), {k: v.clone() for k, v in grads.items()}, stats
    finally:
        sepconv._FOLD = old


CASES = [
    # n, cin, cout, h, w, stride, dil, residual, relu
    (2, 728, 728, 33, 33, 1, 2, True, True),       # middle-flow unit (output stride 8)
    (2, 128, 256, 40, 36, 1, 1, False, True),      # growing unit, Cout = 2 tiles
    (3, 256, 256, 31, 29, 2, 1, False, False),     # strided depthwise, odd map
    (1, 1536, 2048, 17, 17, 1, 4, False, True),    # exit-flow conv5
]


@pytest.mark.parametrize("case", CASES)
def test_folded_unit_matches_fp32_reference(case):
    n, cin, cout, h, w, stride, dil, with_res, relu = case
    torch.manual_seed(3)
    unit = _Unit(cin, cout, stride, dil, relu_after=relu).to(DEV).train()
    with torch.no_grad():
        # non-trivial affine parameters and a depthwise output with per-channel offsets and scales
        unit.sep.bn.weight.uniform_(0.5, 1.5); unit.sep.bn.bias.normal_(0, 0.5)
        unit.bn.weight.uniform_(0.5, 1.5); unit.bn.bias.normal_(0, 0.5)
        unit.sep.conv1.weight.mul_(torch.rand(cin, 1, 1, 1, device=DEV) * 3 + 0.2)
    x = (torch.randn(n, cin, h, w, device=DEV) + torch.randn(1, cin, 1, 1, device=DEV)).to(torch.bfloat16)
    x = x.contiguous(memory_format=torch.channels_last)
    ho = (h + 2 * dil - 2 * dil - 1) // stride + 1
    wo = (w + 2 * dil - 2 * dil - 1) // stride + 1
    gy = torch.randn(n, cout, ho, wo, device=DEV).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    res = None
    if with_res:
        res = torch.randn(n, cout, ho, wo, device=DEV).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)

    assert sepconv.can_fold(unit.sep, unit.bn, x.requires_grad_(True)), "folded path not taken on the GPU box"
    x = x.detach()
    yr, dxr, drr, gr, sr = _reference(unit, x, gy, res, relu, dil, stride)
    yf, dxf, drf, gf, sf = _run_gpu(unit, x, gy, res, relu, fold=True)
    yu, dxu, dru, gu, su = _run_gpu(unit, x, gy, res, relu, fold=False)

    assert yf.shape == yr.shape
    assert _rel(yf, yr) < 1.5e-2 and _cos(yf, yr) > 0.9999
    assert _chan_rel(yf, yr, 1) < 4e-2
    # behind a ReLU the bf16 pipeline flips the mask of the ~0.3 % of outputs that round across zero, each flip a
    # full-size gradient error (measured: the unfolded kernels and the folded ones both land at cos 0.9992,
    # gpurun_out/e/diag.log); without the ReLU the gradients agree to bf16 rounding
    cos_min, rel_max = (0.9985, 0.2) if relu else (0.9999, 1.5e-2)
    assert _cos(dxf, dxr) > cos_min and _rel(dxf, dxr) < rel_max
    if with_res:
        assert _cos(drf, drr) > cos_min
    for k in ("dw", "pw", "g1", "g2", "b2"):
        assert _cos(gf[k], gr[k]) > cos_min, (k, _cos(gf[k], gr[k]))
        assert _rel(gf[k], gr[k]) < rel_max, (k, _rel(gf[k], gr[k]))
    # dbeta1 is zero in exact arithmetic (BN2 cancels a constant); the folded path returns exactly that
    assert float(gf["b1"].abs().max()) == 0.0
    assert float(gr["b1"].abs().max()) <= 1e-3 * max(1.0, float(gr["g1"].abs().max()))
    for k in ("rm1", "rv1", "rm2", "rv2"):
        assert _rel(sf[k], sr[k]) < 2e-3, (k, _rel(sf[k], sr[k]))
    # the folded path is no further from the reference than the unfolded kernels are (x 1.5 + slack)
    assert _rel(yf, yr) <= 1.5 * _rel(yu, yr) + 5e-3
    assert 1 - _cos(dxf, dxr) <= 1.5 * (1 - _cos(dxu, dxr)) + 1e-4
    for k in ("pw", "g1", "dw"):
        assert 1 - _cos(gf[k], gr[k]) <= 1.5 * (1 - _cos(gu[k], gr[k])) + 1e-4, k


def test_folded_unit_with_gradient_arena():
    """Inside the engine the weight / affine gradients go to the arena sinks and the DDP hooks are notified: one
    engine step of a tiny Xception-style stack with and without folding gives the same parameters."""
    from distributedpytorch_amd.models.xception import XceptionBlock
    from distributedpytorch_amd.parallel import arena as arena_mod

    def run(fold, arena):
        old = sepconv._FOLD
        sepconv._FOLD = fold
        try:
            torch.manual_seed(11)
            blk = XceptionBlock(64, 64, 3, dilation=2).to(DEV).train()
            with torch.no_grad():
                # (with the default beta = 0 a unit's output scale is cancelled by the next unit's depthwise BN:
                # d(gamma) would be zero in exact arithmetic and the comparison one of rounding noise)
                for m in blk.modules():
                    if hasattr(m, "running_mean"):
                        m.weight.uniform_(0.5, 1.5); m.bias.normal_(0, 0.7)
            ar = arena_mod.ParamArena([list(blk.named_parameters())], lowp_dtype=torch.bfloat16) if arena else None
            x = torch.randn(4, 64, 24, 24, device=DEV).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
            y = blk(x.requires_grad_(True))
            y.float().square().mean().backward()
            torch.cuda.synchronize()
            return {n: p.grad.detach().float().clone() for n, p in blk.named_parameters()}, ar
        finally:
            sepconv._FOLD = old

    gf, ar = run(True, arena=True)
    assert all(s.param.grad.data_ptr() == s.param._dpamd_grad_sink.data_ptr() for s in ar.slots), \
        "gradients left the arena"
    gu, _ = run(False, arena=False)
    for k in gu:
        if k.endswith("sep.bn.bias"):
            continue                       # zero in exact arithmetic; noise on the unfolded path
        c = _cos(gf[k], gu[k])
        assert c > 0.98, (k, c)


class _TinyXception(torch.nn.Module):
    """stem conv + BN + ReLU -> a strided block with a shortcut convolution -> two identity-shortcut blocks: every
    fused backward path of models.xception (shortcut passthrough through the dense and the depthwise data-gradient
    kernels, BatchNorm backward prologues in both, folded units) in one small network."""

    def __init__(self, stride=1):
        super().__init__()
        from distributedpytorch_amd.models.xception import XceptionBlock
        from distributedpytorch_amd.ops import BatchNorm2d, Conv2d
        self.conv = Conv2d(16, 64, 3, padding=1, bias=False)
        self.bn = BatchNorm2d(64)
        self.b1 = XceptionBlock(64, 128, 2, stride=stride, start_with_relu=False)
        self.b2 = XceptionBlock(128, 128, 3, dilation=2)
        self.b3 = XceptionBlock(128, 128, 3, dilation=1)
        for b in (self.b1, self.b2, self.b3):
            b.activate_output, b.input_sole = True, True
        self.b2.input_activated = self.b3.input_activated = True

    def forward(self, x):
        from distributedpytorch_amd.ops import conv_bn_act
        x = conv_bn_act(self.conv, self.bn, x, relu=True)
        return self.b3(self.b2(self.b1(x)))


@pytest.mark.parametrize("stride", [1, 2])
def test_xception_blocks_match_cpu_fp32(stride):
    torch.manual_seed(5)
    net = _TinyXception(stride).train()
    with torch.no_grad():
        for m in net.modules():
            if hasattr(m, "running_mean"):
                m.weight.uniform_(0.6, 1.4); m.bias.normal_(0, 0.5)
    x = torch.randn(3, 16, 34, 30).to(torch.bfloat16)
    # CPU fp32 reference: the same modules on their torch path, convolution weights rounded to bf16 as on the GPU
    import copy
    ref = copy.deepcopy(net).float()
    with torch.no_grad():
        for p in ref.parameters():
            if p.dim() == 4:
                p.copy_(p.to(torch.bfloat16).float())
    xr = x.float().requires_grad_(True)
    yr = ref(xr)
    gy = torch.randn_like(yr).to(torch.bfloat16)
    yr.backward(gy.float())

    # nine ReLU layers deep, a bf16 pipeline differs from fp32 by the mask flips of the outputs that round across
    # zero: measured cos(dx) 0.974-0.983 and worst parameter cosine 0.92-0.97 for EVERY variant, the fully unfused
    # kernels included (gpurun_out/l/diag.log).  The test is therefore "fused = unfused against the reference", a
    # wiring error shows as a cosine far below either.
    from distributedpytorch_amd.ops import _hip_conv
    res = {}
    for name, fold, prologue in (("fused", True, True), ("plain", False, False)):
        old = sepconv._FOLD, _hip_conv._BN_PROLOGUE
        sepconv._FOLD, _hip_conv._BN_PROLOGUE = fold, prologue
        try:
            g = copy.deepcopy(net).to(DEV)
            xh = x.to(DEV).contiguous(memory_format=torch.channels_last).requires_grad_(True)
            yh = g(xh)
            yh.backward(gy.to(DEV).contiguous(memory_format=torch.channels_last))
            torch.cuda.synchronize()
        finally:
            sepconv._FOLD, _hip_conv._BN_PROLOGUE = old
        worst, allg, allr = 1.0, [], []
        for (k, pg), (_, pr) in zip(g.named_parameters(), ref.named_parameters()):
            if k.endswith("sep.bn.bias") or float(pr.grad.norm()) < 1e-6 * float(pr.norm() + 1):
                continue                 # zero in exact arithmetic (cancelled by the next BatchNorm)
            worst = min(worst, _cos(pg.grad, pr.grad))
            allg.append(pg.grad.detach().float().cpu().flatten()); allr.append(pr.grad.detach().float().flatten())
        overall = _cos(torch.cat(allg), torch.cat(allr))
        stats_err = max(_rel(bg, br) for (k, bg), (_, br) in zip(g.named_buffers(), ref.named_buffers())
                        if k.endswith("running_mean") or k.endswith("running_var"))
        res[name] = (_cos(yh, yr), _cos(xh.grad, xr.grad), worst, stats_err, overall)
    print(res)
    yf, dxf, wf, sf, of = res["fused"]
    yp, dxp, wp, sp, op = res["plain"]
    # The WORST single parameter's cosine moves between runs of one binary (0.89-0.96 fused, 0.94-0.97 plain over
    # six runs, gpurun_out/ae + af: the weight gradients are summed by fp32 atomics in launch order, and a sum that
    # rounds differently flips ReLU masks downstream), so it only gets a floor; the comparison between the
    # variants is made on the cosine over ALL parameter gradients together, which the large tensors stabilise.
    assert yf > 0.9995 and dxf > 0.95 and wf > 0.85 and sf < 2e-2, res
    assert of > 0.95 and of > op - 0.02 and dxf > dxp - 0.02, res
