"""``relu(BN(conv1x1(a)) + identity)`` — the tail of a ResNet bottleneck — with the BatchNorm's backward folded
algebraically into the 1x1 convolution's two gradients.

SURVEY.md §2.5 K4/K5 (BN+residual+ReLU) around K1-K3; the reference reaches the bottleneck only through its
external model (`/root/reference/train_pascal.py:86`).  Why: the headline step is bound by HBM traffic in its
1x1 / BatchNorm half (tools/overlap_probe.py: the 256 -> 1024 weight gradient streams at 3.9 TB/s, bn_bwd_apply at
4.6 TB/s), and the widest tensors of a bottleneck are the 4U-wide ones around its last BatchNorm (U = one
[M, planes] bf16 tensor).  The plain backward moves, per block,

    consumer's dgrad store loop   reads y3 (4U) for Σ dm·x̂            4U
    bn_bwd_apply                  reads dm, y3, writes dy3             12U
    conv3 dgrad / wgrad           read dy3                              8U (+ 3U of a2 / x2 / output)

BatchNorm backward is LINEAR in its input y3 = W a2 (W: [Cout, Cin] of the 1x1 convolution):

    dy3 = ka * dm + kb * y3 + kc         ka = g rstd,  kb = -ka rstd Σ(dm x̂) / M,  kc = -ka Σdm / M - kb mean

so with T = dmᵀ a2 (the weight-gradient kernel on the MASKED gradient dm, which the consumer's store loop leaves
behind anyway), A = a2ᵀ a2, s = Σ_p a2:

    Σ dm x̂  = rstd (Σ_j W_kj T_kj - mean Σdm)                         (no read of y3 anywhere)
    dW      = ka T + kb (W A) + kc sᵀ
    da2     = (diag(ka) W)ᵀ dm + G a2 + c0,      G = Wᵀ diag(kb) W  [Cin x Cin],   c0 = Wᵀ kc

dy3 never exists and y3 is not kept for backward (4U of memory per block).  Traffic: weight gradient 5U, A and s
2U, t = G a2 + c0 2U, data gradient 7U = 16U against 27U, and the consumer's store loop reads 4U less.

Synchronised statistics: Σdm and Σ dm·x̂ are all-reduced (one [2, C] message) before ka / kb / kc are formed; A, s
and T stay local (dW is a local gradient, averaged by DDP like every other).
"""
from __future__ import annotations

import os
from typing import Optional

import torch
import torch.distributed as dist

from . import _backend as B

_ENABLED = os.environ.get("DPAMD_BN3_FOLD", "1") != "0"
COUNTS = {"folded": 0, "fallback": 0}      # backward passes that took the folded / the recomputing path


def _sink_of(p):
    s = getattr(p, "_dpamd_grad_sink", None)
    if s is not None and p.grad is not None and p.grad.data_ptr() == s.data_ptr():
        return s
    return None


def _notify(p):
    fn = getattr(p, "_dpamd_on_grad", None)
    if fn is not None and p.requires_grad:
        fn(p)


class _ConvBnAddReluFold(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, identity, weight, gamma, beta, w_lowp, bn, in_holder, out_holder, group):
        from . import _hip_bn, _hip_conv
        from .batchnorm import _group_active, _small_all_reduce
        a = _hip_bn.as_nhwc(a)
        identity = _hip_bn.as_nhwc(identity)
        n, cin, h, w = a.shape
        cout = weight.shape[0]
        m_rows = n * h * w
        reps = _hip_conv.stat_replicas(m_rows)
        stats = _hip_conv.stats_buffer(reps, cout, a.device)
        y3 = _hip_conv.igemm(a, w_lowp, (h, w), 1, 1, (0, 0), (1, 1), None, stats)
        stats = _hip_conv.fold_stats(stats)
        sync = _group_active(group)
        pg = None if group is True else group
        count = float(m_rows)
        if sync:
            from ..parallel import p2p
            if pg is None and p2p.resolve() and stats.numel() > p2p._MAXN:
                stats = stats.sum(0, keepdim=True)
            _small_all_reduce(stats.view(-1), pg)
            count *= dist.get_world_size(pg)
        g32, b32 = gamma.detach().float().contiguous(), beta.detach().float().contiguous()
        bits = torch.empty(y3.numel() // 8, dtype=torch.uint8, device=a.device)
        out, mean, invstd = _hip_bn.apply_from_stats(y3, None, None, g32, b32, identity, bn.running_mean,
                                                     bn.running_var, count, bn.eps, bn.momentum, True, stats,
                                                     mask_out=bits)
        del y3                               # not kept: backward never reads it (module docstring)
        # what the consumer's data-gradient store loop needs: the mask; Σ dm·x̂ comes out of T (x = None)
        out_holder.update(x=None, mask_from_a=True, mask_bits=bits, fold=True, mean=None, invstd=None, w=None, b=None)
        ctx.save_for_backward(a, w_lowp, mean, invstd, g32, b32, out)
        ctx.params = (weight, gamma, beta)
        ctx.in_holder = in_holder if (in_holder and "x" in in_holder and in_holder.get("x") is not None) else None
        ctx.out_holder = out_holder
        ctx.count, ctx.sync, ctx.pg = count, sync, pg
        ctx.set_materialize_grads(False)
        return out

    @staticmethod
    def backward(ctx, dy):
        if dy is None:
            return (None,) * 10
        from . import _hip_bn, _hip_conv, workspace
        from .batchnorm import _small_all_reduce
        a, w_lowp, mean, invstd, g32, b32, out = ctx.saved_tensors
        weight, gamma, beta = ctx.params
        n, cin, h, w = a.shape
        cout = w_lowp.shape[0]
        dev = a.device
        dy = _hip_bn.as_nhwc(dy)
        hold = ctx.out_holder
        cl = torch.channels_last
        sink_scale = 1.0 / dist.get_world_size(ctx.pg) if ctx.sync else 1.0
        w_sink, g_sink, b_sink = _sink_of(weight), _sink_of(gamma), _sink_of(beta)
        fused = hold.get("dm_ptr") == dy.data_ptr() and hold.get("sums") is not None
        dw = dgamma = dbeta = None
        COUNTS["folded" if fused else "fallback"] += 1
        if not fused:
            # the consumer did not run the mask + Σdm prologue (several consumers, an unfused shape): the ordinary
            # composition on a recomputed y3
            y3 = _hip_conv.igemm(a, w_lowp, (h, w), 1, 1, (0, 0), (1, 1))
            sdy, sdyx = _hip_bn.bwd_reduce(dy, y3, out, mean, invstd, g32, b32, 1)
            lsdy, lsdyx = sdy, sdyx
            if ctx.sync:
                packed = torch.stack([sdy, sdyx])
                _small_all_reduce(packed, ctx.pg)
                sdy, sdyx = packed[0], packed[1]
            dy3, dres = _hip_bn.bwd_apply(dy, y3, out, mean, invstd, g32, b32, sdy, sdyx, ctx.count, 1, True)
            del y3
            dgam, dbet = (sdyx, sdy) if ctx.sync else (lsdyx, lsdy)
            wt = _hip_conv.weight_transpose(w_lowp)
            da = None
            if ctx.needs_input_grad[0]:
                f = _hip_conv.dgrad_bn_fused(dy3, wt, (h, w), 1, (0, 0), (1, 1), None, ctx.in_holder, a) \
                    if ctx.in_holder is not None else None
                if f is not None:
                    da, part = f
                    ctx.in_holder["sums"], ctx.in_holder["dm_ptr"] = part, da.data_ptr()
                else:
                    da = _hip_conv.igemm(dy3, wt, (h, w), 1, 1, (0, 0), (1, 1))
            if w_sink is not None:
                _hip_conv.wgrad(dy3, a, (cout, cin, 1, 1), (1, 1), (0, 0), (1, 1), out=w_sink)
            else:
                dw = _hip_conv.wgrad(dy3, a, (cout, cin, 1, 1), (1, 1), (0, 0), (1, 1))
        else:
            part = hold["sums"]                                   # fp32 [reps, 2, C]: (Σdm, unused)
            hold["sums"] = None
            dres = dy                                             # the identity's gradient IS dm (no copy)
            # T = dmᵀ a2, A = a2ᵀ a2 (weight-gradient kernel), s = Σ_p a2 (statistics kernel)
            T = workspace.zeros(cout * cin, dev).view(cout, cin, 1, 1)
            _hip_conv.wgrad(dy, a, (cout, cin, 1, 1), (1, 1), (0, 0), (1, 1), out=T)
            A = workspace.zeros(cin * cin, dev).view(cin, cin, 1, 1)
            _hip_conv.wgrad(a, a, (cin, cin, 1, 1), (1, 1), (0, 0), (1, 1), out=A)
            s, _ = _hip_bn.stats(a)
            # Σdm, Σ dm·x̂ (csrc/bn_relu.hip, bn3fold_*); under SyncBN the global sums
            sums = _hip_bn.bn3fold_rows(w_lowp, T.view(cout, cin), part, mean, invstd)
            if ctx.sync:
                _small_all_reduce(sums, ctx.pg)
            if g_sink is None or b_sink is None:
                dgamma32 = torch.zeros(cout, dtype=torch.float32, device=dev)
                dbeta32 = torch.zeros(cout, dtype=torch.float32, device=dev)
            wf, kbw, kvec = _hip_bn.bn3fold_coef(w_lowp, sums, sums, mean, invstd, g32, ctx.count, sink_scale,
                                                 g_sink if g_sink is not None and b_sink is not None else dgamma32,
                                                 b_sink if g_sink is not None and b_sink is not None else dbeta32)
            if g_sink is None or b_sink is None:
                dgamma, dbeta = dgamma32.to(gamma.dtype), dbeta32.to(beta.dtype)
            WA = wf @ A.view(cin, cin)                            # two small fp32 library GEMMs
            G = wf.t() @ kbw                                      # [K1, K1 + 8]: G | c0 | 0
            wp = weight
            wT = getattr(wp, "_dpamd_lowpT", None)
            if not (wT is not None and getattr(wp, "_dpamd_lowpT_ok", False)
                    and w_lowp.data_ptr() == getattr(wp, "_dpamd_lowp", w_lowp).data_ptr()
                    and getattr(wp, "_dpamd_lowp", None) is not None):
                wT = _hip_conv.weight_transpose(w_lowp)
            if w_sink is None:
                dw = torch.zeros((cout, cin, 1, 1), dtype=torch.float32, device=dev).contiguous(memory_format=cl)
            gb, c0, wpt = _hip_bn.bn3fold_comb(T, WA, s, kvec, wf, G, w_sink if w_sink is not None else dw, wT)
            da = None
            if ctx.needs_input_grad[0]:
                t = _hip_conv.igemm(a, gb, (h, w), 1, 1, (0, 0), (1, 1), bias=c0)
                f = _hip_conv.dgrad_bn_fused(dy, wpt, (h, w), 1, (0, 0), (1, 1), t, ctx.in_holder, a) \
                    if ctx.in_holder is not None else None
                if f is not None:
                    da, part2 = f
                    ctx.in_holder["sums"], ctx.in_holder["dm_ptr"] = part2, da.data_ptr()
                else:
                    da = _hip_conv.igemm(dy, wpt, (h, w), 1, 1, (0, 0), (1, 1), residual=t)
            for prm, grad_here in ((weight, dw), (gamma, dgamma), (beta, dbeta)):
                if grad_here is None:
                    _notify(prm)
            if not ctx.needs_input_grad[1]:
                dres = None
            return da, dres, dw, dgamma, dbeta, None, None, None, None, None
        # affine gradients (local sums; under SyncBN the global ones scaled by 1 / world: the same after DDP's average)
        if g_sink is not None and b_sink is not None:
            g_sink.add_(dgam, alpha=sink_scale)
            b_sink.add_(dbet, alpha=sink_scale)
        else:
            dgamma = (dgam * sink_scale).to(gamma.dtype)
            dbeta = (dbet * sink_scale).to(beta.dtype)
        for prm, grad_here in ((weight, dw), (gamma, dgamma), (beta, dbeta)):
            if grad_here is None:
                _notify(prm)
        if not ctx.needs_input_grad[1]:
            dres = None
        return da, dres, dw, dgamma, dbeta, None, None, None, None, None


def can_fold(conv, bn, x: torch.Tensor, residual: Optional[torch.Tensor]) -> bool:
    return (_ENABLED and B.use_hip(x) and x.dtype == torch.bfloat16 and torch.is_grad_enabled()
            and residual is not None and bn.training and not bn.frozen and bn.weight is not None
            and conv.kernel_size == (1, 1) and conv.stride == (1, 1) and conv.padding == (0, 0)
            and conv.groups == 1 and conv.bias is None and conv.in_channels % 8 == 0 and conv.out_channels % 8 == 0
            and tuple(residual.shape) == (x.shape[0], conv.out_channels, x.shape[2], x.shape[3])
            and residual.dtype == torch.bfloat16)


def conv_bn_add_relu(conv, bn, x: torch.Tensor, residual: torch.Tensor, sole_consumer: bool = True):
    """``relu(BN(conv(x)) + residual)`` for a 1x1 ``ops.Conv2d`` and an ``ops.BatchNorm2d``; the folded backward
    where it applies (GPU training path, batch statistics), the ordinary composition elsewhere.  The result carries
    the holder its consumer's data-gradient kernel reads (mask bits; no pre-BN tensor)."""
    from .conv import conv_bn_act
    if not can_fold(conv, bn, x, residual):
        return conv_bn_act(conv, bn, x, relu=True, residual=residual, sole_consumer=sole_consumer)
    in_holder = getattr(x, "_dpamd_bn_holder", None) if sole_consumer else None
    from . import _hip_conv
    if not _hip_conv._BN_PROLOGUE:
        in_holder = None
    group = False
    if bn.sync:
        group = bn.process_group if bn.process_group is not None else True
    out_holder: dict = {}
    y = _ConvBnAddReluFold.apply(x, residual, conv.weight, bn.weight, bn.bias, conv.lowp_weight(x.dtype), bn,
                                 in_holder, out_holder, group)
    y._dpamd_bn_holder = out_holder
    return y
