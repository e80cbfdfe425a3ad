"""Separable-convolution unit ``depthwise 3x3 -> BN -> pointwise 1x1 -> BN (+residual)(+ReLU)`` with the inner
BatchNorm folded into the pointwise weights.

SURVEY.md §2.5 K13 / BASELINE.json config 5 (DeepLabv3+ Xception-65, output stride 8).  The reference reaches this
unit only through its external model (`/root/reference/train_pascal.py:86`: ``backbone='xception'``); what it
lowers to on MI355X is decided here.

Xception's separable unit has NO activation between the depthwise BatchNorm and the pointwise convolution, so

    pointwise(BN1(z)) = W (s * z + t) = (W diag(s)) z + W t,      s = gamma1 * rstd1,  t = beta1 - mean1 * s

and the constant ``W t`` is cancelled by the BatchNorm that follows.  The folded path therefore never writes or
reads the normalised tensor: the depthwise kernel's epilogue leaves the statistics of ``z``, a [Cout x Cin] kernel
scales the weights, the MFMA implicit-GEMM consumes the raw ``z``.  Per unit and direction that removes one full
read+write pass over the (728-channel, stride-8) activation in forward and two in backward (the measured profile
of the unfused path: profiles/r2_x65_os8_b64_v28_profile.md, BatchNorm kernels 121 of 307 ms).

Synchronised statistics (``bn1.sync``): the partial sums of z are all-reduced before the fold (forward) and ds
before the coefficient c is formed (backward); dW stays a local gradient.

Backward, with dW' the weight gradient w.r.t. the folded weights (computed by the ordinary weight-gradient kernel
on the raw z) and M = N*H*W:

    dW      = dW' diag(s)
    ds_j    = sum_k dW'_kj W_kj            dgamma1_j = ds_j * rstd_j          dbeta1 = 0 (exactly: BN2 cancels it)
    dz      = W'^T dy + c * (z - mean1),    c_j = -ds_j * gamma1_j * rstd_j^3 / M

(the second term is the gradient through rstd1; the gradient through mean1 vanishes for the same reason dbeta1
does).  The following BatchNorm's running mean is corrected for the constant it does not see.
"""
from __future__ import annotations

import os
from typing import Optional

import torch

from . import _backend as B
from .conv import conv_bn_act

_FOLD = os.environ.get("DPAMD_SEPCONV_FOLD", "1") != "0"
# backward schedule of the folded unit: "1" = weight gradient -> data gradient with the statistics term in its
# store loop (one pass less over the activation), "0" = weight and data gradient side by side on two streams and
# a separate elementwise pass for the statistics term
_SERIAL = True


def _sink_of(p):
    s = getattr(p, "_dpamd_grad_sink", None)
    if s is not None and p.grad is not None and p.grad.data_ptr() == s.data_ptr():
        return s
    return None


class _FoldedPointwise(torch.autograd.Function):
    """y', stats(y') = (W diag(s)) z with s from the statistics of z.  Inputs 1-3 are differentiable."""

    @staticmethod
    def forward(ctx, z, gamma, beta, weight, stats1, w_lowp, bn1, bn2, group=False):
        from . import _hip_bn, _hip_conv
        from .batchnorm import _group_active, _small_all_reduce
        n, cin, h, w = z.shape
        cout = weight.shape[0]
        count = float(n * h * w)
        # synchronised statistics (bn1.sync): the partial sums of z are all-reduced in place — one message — and the
        # fold runs on the global mean / variance; backward sums ds over the ranks the same way
        ctx.sync = _group_active(group)
        ctx.pg = None if group is True else group
        if ctx.sync:
            import torch.distributed as dist
            _small_all_reduce(stats1.view(-1), ctx.pg)
            count *= dist.get_world_size(ctx.pg)
        g32, b32 = gamma.detach().float().contiguous(), beta.detach().float().contiguous()
        need_t = bool(ctx.needs_input_grad[0])
        wf, vec, _, *wft = _hip_bn.fold_into_pointwise(stats1, g32, b32, bn1.running_mean, bn1.running_var, count,
                                                       bn1.eps, bn1.momentum, w_lowp, bn2.running_mean, bn2.momentum,
                                                       transposed=need_t)
        ctx.wft = wft[0] if wft else None         # (not a saved tensor: never an input or output of this node)
        reps = _hip_conv.stat_replicas(n * h * w)
        stats2 = _hip_conv.stats_buffer(reps, cout, z.device)
        y = _hip_conv.igemm(z, wf, (h, w), 1, 1, (0, 0), (1, 1), None, stats2)
        stats2 = _hip_conv.fold_stats(stats2)
        ctx.save_for_backward(z, wf, w_lowp, vec, g32)
        ctx.params = (gamma, beta, weight)
        ctx.count = count
        ctx.set_materialize_grads(False)
        ctx.mark_non_differentiable(stats2)
        return y, stats2

    @staticmethod
    def backward(ctx, dy, _unused=None):
        if dy is None:
            return (None,) * 9
        from . import _hip_bn, _hip_conv, streams, workspace
        from .batchnorm import _small_all_reduce
        z, wf, w_lowp, vec, g32 = ctx.saved_tensors
        wft = ctx.wft if ctx.wft is not None else _hip_conv.weight_transpose(wf)
        gamma, beta, weight = ctx.params
        cout, cin = wf.shape[:2]
        n, _, h, w = z.shape
        dy = _hip_bn.as_nhwc(dy)
        # 1. weight gradient w.r.t. the folded weights (needed first: ds, hence the data gradient's statistics
        #    term, comes out of it)
        # (a slice of the per-step zeroed pool — one memset per step — not a fill launch per layer)
        dwf = workspace.zeros(cout * cin, z.device).view(cout, 1, 1, cin).permute(0, 3, 1, 2)
        serial = _SERIAL or not streams.enabled()
        aux = None
        if serial:
            _hip_conv.wgrad(dy, z, (cout, cin, 1, 1), (1, 1), (0, 0), (1, 1), out=dwf)
        else:
            aux = streams.fork(z.device)
            with torch.cuda.stream(aux):
                _hip_conv.wgrad(dy, z, (cout, cin, 1, 1), (1, 1), (0, 0), (1, 1), out=dwf)
            streams.keep_alive(dy, z)
        need_dz = ctx.needs_input_grad[0]
        dz = None
        if need_dz and not serial:
            # data gradient beside the weight gradient; the statistics term is a separate pass behind both
            dz = _hip_conv.igemm(dy, wft, (h, w), 1, 1, (0, 0), (1, 1))
        if aux is not None:
            torch.cuda.current_stream(z.device).wait_stream(aux)
        w_sink = _sink_of(weight) if ctx.needs_input_grad[3] else None
        g_sink = _sink_of(gamma) if ctx.needs_input_grad[1] else None
        dw = dgamma = dbeta = None
        if w_sink is None:
            dw = torch.zeros_like(dwf)
        if g_sink is None:
            dgamma = torch.zeros(cin, dtype=torch.float32, device=z.device)
        sink_scale, reduce_ds = 1.0, None
        if ctx.sync:
            import torch.distributed as dist
            sink_scale = 1.0 / dist.get_world_size(ctx.pg)
            reduce_ds = (lambda t: _small_all_reduce(t, ctx.pg))
        cc = _hip_bn.fold_backward(dwf, w_lowp, vec, g32, w_sink if w_sink is not None else dw,
                                   g_sink if g_sink is not None else dgamma, ctx.count, sink_scale, reduce_ds)
        if need_dz and serial:
            # 2. dz = W'^T dy + c * z - c * mean in ONE kernel: the statistics term rides in the store loop
            dz = _hip_conv.igemm(dy, wft, (h, w), 1, 1, (0, 0), (1, 1),
                                 residual=z, residual_affine=cc)
        elif need_dz:
            _hip_bn.affine_add_(dz, z, cc[0], cc[1])
        if ctx.needs_input_grad[2] and _sink_of(beta) is None:
            dbeta = torch.zeros(cin, dtype=beta.dtype, device=z.device)
        if dgamma is not None:
            dgamma = dgamma.to(gamma.dtype) if ctx.needs_input_grad[1] else None
        if dw is not None and not ctx.needs_input_grad[3]:
            dw = None
        for prm, grad_here in ((weight, dw), (gamma, dgamma), (beta, dbeta)):
            if grad_here is None:
                notify = getattr(prm, "_dpamd_on_grad", None)
                if notify is not None and prm.requires_grad:
                    notify(prm)
        return dz, dgamma, dbeta, dw, None, None, None, None, None


def can_fold(sep, bn2, x: torch.Tensor) -> bool:
    dw, bn1, pw = sep.conv1, sep.bn, sep.pointwise
    return (_FOLD and B.use_hip(x) and x.dtype == torch.bfloat16 and torch.is_grad_enabled()
            and bn1.training and not bn1.frozen
            and bn2.training and not bn2.frozen and 0.0 <= bn2.momentum < 1.0
            and dw.can_fuse_stats(x) and pw.bias is None and pw.groups == 1 and pw.kernel_size == (1, 1)
            and pw.stride == (1, 1) and pw.padding == (0, 0)
            and pw.in_channels % 8 == 0 and pw.out_channels % 8 == 0 and pw.in_channels <= 8192)


EVAL_FOLD = True          # (module switch for tests / tools)


def _eval_fold_ok(sep, bn2, x: torch.Tensor) -> bool:
    pw = sep.pointwise
    return (EVAL_FOLD and not bn2.training and not sep.bn.training and not torch.is_grad_enabled() and B.use_hip(x)
            and x.dtype == torch.bfloat16 and pw.bias is None and pw.groups == 1 and pw.kernel_size == (1, 1)
            and pw.stride == (1, 1) and pw.padding == (0, 0) and pw.in_channels % 8 == 0 and pw.out_channels % 8 == 0)


@torch.no_grad()
def eval_fold_constants(sep, bn2=None):
    """Evaluation constants of a separable unit: with running statistics BN1 is a per-channel affine map between two
    linear maps, so ``BN2(pointwise(BN1(z))) = (W diag(s1)) z * s2 + (s2 * (W t1) + t2)`` — the pointwise weights with
    BN1's scale folded in (bf16, the layout of the kernels' weights) and one (scale | shift) block for the
    convolution's evaluation epilogue; the pass of BN1 over the depthwise output is gone.  Computed once per
    evaluation phase INTO FIXED STORAGE (captured graphs keep reading it; ``parallel.graphs`` calls
    ``sep.eval_refresh()`` before a replay); stale after what makes ``BatchNorm2d.eval_affine`` stale."""
    st = sep.__dict__.get("_eval_fold")
    if bn2 is None:
        if st is None:
            return None
        bn2 = st["bn2"]
    pw, bn1 = sep.pointwise, sep.bn
    a1, a2 = bn1.eval_affine(), bn2.eval_affine()
    key = (pw.weight._version, bn1._affine_key, bn2._affine_key, sep.__dict__.get("_eval_fold_epoch", 0))
    dev = pw.weight.device
    if st is None or st["bn2"] is not bn2 or st["w"].device != dev:
        st = {"bn2": bn2, "key": None,
              "w": torch.empty(pw.weight.shape, dtype=torch.bfloat16, device=dev).contiguous(memory_format=torch.channels_last),
              "aff": torch.empty(2, pw.out_channels, dtype=torch.float32, device=dev)}
        sep.__dict__["_eval_fold"] = st
    if st["key"] != key:
        w = pw.weight.detach().float()
        st["w"].copy_(w * a1[0].view(1, -1, 1, 1))
        st["aff"][0].copy_(a2[0])
        st["aff"][1].copy_(a2[0] * (w.flatten(1) @ a1[1]) + a2[1])
        st["key"] = key
    return st


def sepconv_bn_act(sep, bn2, x: torch.Tensor, relu: bool = False, residual: Optional[torch.Tensor] = None,
                   dropout: float = 0.0, passthrough: bool = False, sole_consumer: bool = False):
    """``act(BN2(pointwise(BN1(depthwise(x)))) + residual)`` for a ``models.xception.SeparableConv2d`` ``sep``.
    Training on the GPU path with local (non-synchronised, non-frozen) statistics folds BN1 into the pointwise
    weights; everything else runs the ordinary composition.

    ``passthrough=True`` returns ``(out, x')``: ``x'`` is ``x`` routed through the depthwise node — use it as the
    block's shortcut and its gradient is added inside the depthwise data-gradient kernel (no add kernel).
    ``sole_consumer``: ``x`` (a fused BN+ReLU output) feeds nothing but this unit (and its passthrough), so that
    BatchNorm's backward reductions run in the same kernel's epilogue."""
    dw = sep.conv1
    if _eval_fold_ok(sep, bn2, x):
        from . import _hip_conv
        st = eval_fold_constants(sep, bn2)
        z = _hip_conv.as_nhwc(dw(x))
        res = _hip_conv.as_nhwc(residual) if residual is not None else None
        y = _hip_conv.igemm_infer(z, st["w"], tuple(z.shape[2:]), 1, 1, (0, 0), (1, 1), st["aff"], relu, res)
        if y is None:            # (a shape the kernel does not take: BN1 as its own pass after all)
            y = conv_bn_act(sep.pointwise, bn2, sep.bn(z), relu=relu, residual=residual)
        return (y, x) if passthrough else y
    pt = passthrough and dw.can_passthrough(x)
    if not can_fold(sep, bn2, x):
        if bn2.training and not sep.bn.frozen and dw.can_fuse_stats(x) and (pt or sole_consumer) and sep.bn.training:
            if pt:
                z, st, skip = dw(x, with_stats=True, passthrough=True, sole_consumer=sole_consumer)
            else:
                z, st = dw(x, with_stats=True, sole_consumer=sole_consumer)
            a = sep.bn(z, stats=st)
        else:
            pt = False
            a = sep.depthwise_bn(x)
        out = conv_bn_act(sep.pointwise, bn2, a, relu=relu, residual=residual, dropout=dropout)
        return (out, skip if pt else x) if passthrough else out
    if pt:
        z, stats1, skip = dw(x, with_stats=True, passthrough=True, sole_consumer=sole_consumer)
    else:
        z, stats1 = dw(x, with_stats=True, sole_consumer=sole_consumer)
    bn1, pw = sep.bn, sep.pointwise
    group = False
    if bn1.sync:
        group = bn1.process_group if bn1.process_group is not None else True
    y, stats2 = _FoldedPointwise.apply(z, bn1.weight, bn1.bias, pw.weight, stats1, pw.lowp_weight(x.dtype), bn1, bn2,
                                       group)
    out = bn2(y, relu=relu, residual=residual, stats=stats2, dropout=dropout)
    return (out, skip if pt else x) if passthrough else out
