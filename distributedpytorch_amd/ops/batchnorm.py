"""Fused BatchNorm (+residual) (+ReLU) with optional cross-rank statistics synchronisation.

SURVEY.md §2.5 K4/K5 (fused BN-stats, BN-apply+ReLU, fused BN+ReLU backward) and §2.6 C4
(SyncBatchNorm all-reduce of ``[Σx, Σx², count]`` forward and ``[Σdy, Σdy·x̂]`` backward).  The
reference only names the knob (``sync_bn=False, freeze_bn=True`` in the commented DeepLab ctor,
`/root/reference/train_pascal.py:85`); semantics follow ``torch.nn.BatchNorm2d``/``SyncBatchNorm``.

One ``autograd.Function`` orchestrates four primitives; each primitive is a HIP kernel on GPU and
plain tensor math on CPU, so the gloo plumbing test exercises exactly the same collective
schedule as the RCCL run:

    stats      x            -> Σx, Σx²                       (per channel, fp32)
    apply      x,scale,shift[,residual] -> y = act(x·scale+shift (+residual))
    bwd_reduce dy,x|y       -> Σdy', Σdy'·x̂                  (dy' = dy masked by the ReLU)
    bwd_apply  dy,x|y,...   -> dx (, d_residual)
"""
from __future__ import annotations

from typing import Optional

import torch
import torch.distributed as dist
import torch.nn as nn
import torch.nn.functional as F

import os

from . import _backend as B

_FUSE_DROPOUT = True
# False: residual blocks keep no positivity bit mask (the consumer re-reads the activation)
_MASK_BITS = True


# --------------------------------------------------------------------------------------
# primitives (torch math = CPU path and numerical reference)
# --------------------------------------------------------------------------------------
def _f(t: torch.Tensor) -> torch.Tensor:
    """fp32 working precision — or fp64 when the caller runs the torch path in double (tools/diag_nondeterminism.py:
    the fp64 reference the kernels' gradients are judged against)."""
    return t if t.dtype == torch.float64 else t.float()


def _t_stats(x: torch.Tensor):
    xf = _f(x)
    return xf.sum(dim=(0, 2, 3)), (xf * xf).sum(dim=(0, 2, 3))


def _t_apply(x, scale, shift, residual, relu):
    y = _f(x) * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1)
    if residual is not None:
        y = y + _f(residual)
    if relu:
        y = torch.relu(y)
    return y.to(x.dtype)


def _t_bwd_reduce(dy, x, y, mean, invstd, relu):
    dyf = _f(dy)
    if relu:
        dyf = dyf * (y > 0)
    xhat = (_f(x) - mean.view(1, -1, 1, 1)) * invstd.view(1, -1, 1, 1)
    return dyf.sum(dim=(0, 2, 3)), (dyf * xhat).sum(dim=(0, 2, 3))


def _t_bwd_apply(dy, x, y, mean, invstd, weight, sum_dy, sum_dy_xhat, count, relu, need_dres):
    dyf = _f(dy)
    if relu:
        dyf = dyf * (y > 0)
    v = lambda t: t.view(1, -1, 1, 1)  # noqa: E731
    xhat = (_f(x) - v(mean)) * v(invstd)
    dx = v(weight * invstd) * (dyf - v(sum_dy) / count - xhat * v(sum_dy_xhat) / count)
    dres = dyf.to(dy.dtype) if need_dres else None
    return dx.to(x.dtype), dres


def _apply_affine(x, scale, shift, residual, relu, out=None):
    if B.use_hip(x):
        from . import _hip_bn
        return _hip_bn.apply(x, scale, shift, residual, relu, out)
    return _t_apply(x, scale, shift, residual, relu)


def _small_all_reduce(packed: torch.Tensor, pg) -> None:
    """SUM of a [2, C] fp32 statistics block over the ranks: the one-shot xGMI peer-to-peer kernel
    (parallel.p2p, DPAMD_SYNCBN_P2P=1, default group, GPU) or the backend's all-reduce."""
    if packed.is_cuda and pg is None:
        from ..parallel import p2p
        if p2p.resolve() and packed.numel() <= p2p._MAXN and packed.is_contiguous():
            p2p.default_instance().all_reduce_(packed)
            return
    if packed.is_cuda:
        from ..parallel.dist_env import all_reduce_
        all_reduce_(packed, pg)
        return
    dist.all_reduce(packed, op=dist.ReduceOp.SUM, group=pg)


# DPAMD_DDP_FORCE=1 (a forced ONE-rank group, parallel.ddp) with --sync-bn: the statistics collectives run too — the
# launch choreography of synchronised BatchNorm measured on a single GPU (timing only: the sums are unchanged)
_FORCE_SYNC = os.environ.get("DPAMD_DDP_FORCE", "0") != "0"


def _group_active(group) -> bool:
    return (group is not False and dist.is_available() and dist.is_initialized()
            and dist.get_world_size(group if group is not True else None) > (0 if _FORCE_SYNC else 1))


# --------------------------------------------------------------------------------------
# training-mode function
# --------------------------------------------------------------------------------------
class _BatchNormActTrain(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, residual, running_mean, running_var, momentum, eps, relu, group,
                conv_stats=None, holder=None, dropout=None, out=None):
        """``dropout`` = (thr16, seed) or None: dropout fused behind the ReLU (HIP path, no residual).
        ``out``: an ops.concat slot — the result is written into that channel slice (HIP path)."""
        C = x.shape[1]
        local_count = x.numel() // C
        hip = B.use_hip(x)
        sync = _group_active(group)
        replicas = None
        if hip:
            from . import _hip_bn
            x = _hip_bn.as_nhwc(x)
            if conv_stats is not None and not sync:
                replicas, s, ss = conv_stats, None, None      # summed inside the apply kernel
            elif conv_stats is not None:
                from ..parallel import p2p
                pg_ = None if group is True else group
                if pg_ is None and p2p.resolve() and conv_stats.numel() > p2p._MAXN:
                    red = conv_stats.sum(0)                   # too long for the one-shot kernel's slots
                    s, ss = red[0], red[1]
                else:
                    # the replicated partial sums are all-reduced IN PLACE as one message and the apply
                    # kernel folds the replicas as on the non-synchronised path: no sum / copy kernels
                    replicas, s, ss = conv_stats, None, None
            else:
                s, ss = _hip_bn.stats(x)
        else:
            s, ss = _t_stats(x)
        count = float(local_count)
        pg = None if group is True else group
        if sync and x.is_cuda and replicas is not None:
            _small_all_reduce(replicas.view(-1), pg)
            count = float(local_count * dist.get_world_size(pg))
            from ..parallel import debug
            if debug.ENABLED:
                debug.check_equal_batches(local_count, x.device)
        elif sync and x.is_cuda:
            # C4 on the GPU: ONE message [Σx | Σx²] per layer and nothing else on the stream — the
            # [2, C] block the conv epilogue (or the statistics kernel) produced is all-reduced in
            # place (no cat / slice / copy kernels, no host round trip for the count: the sampler drops
            # the ragged tail, so every rank contributes the same B·H·W).
            if s.data_ptr() + 4 * C == ss.data_ptr():
                packed = torch.as_strided(s, (2, C), (C, 1))
            else:
                packed = torch.stack([s, ss])
            _small_all_reduce(packed, pg)
            s, ss = packed[0], packed[1]
            count = float(local_count * dist.get_world_size(pg))
            from ..parallel import debug
            if debug.ENABLED:
                debug.check_equal_batches(local_count, x.device)
        elif sync:
            # CPU/gloo: one fused message [Σx | Σx² | count] per layer (2C+1 floats), ragged batches allowed
            packed = torch.cat([s, ss, s.new_tensor([count])])
            dist.all_reduce(packed, op=dist.ReduceOp.SUM, group=pg)
            s, ss = packed[:C].contiguous(), packed[C:2 * C].contiguous()
            count = float(packed[2 * C].item())
        if hip:
            # the apply kernel finalises mean / invstd / running statistics itself (block 0)
            w = weight.detach().float().contiguous() if weight is not None else None
            b = bias.detach().float().contiguous() if bias is not None else None
            mask_mode = 0 if not relu else (1 if residual is not None else 2)
            bits = None
            if mask_mode == 1 and holder is not None and dropout is None and out is None and _MASK_BITS:
                # a residual block's output: its ReLU mask cannot be recomputed from the BN input alone, and the
                # consumer's data-gradient kernel would re-read the whole activation for it — leave 1 bit per
                # element behind instead (csrc/bn_relu.hip, ApplyArgs::mask)
                bits = torch.empty(x.numel() // 8, dtype=torch.uint8, device=x.device)
            y, mean, invstd = _hip_bn.apply_from_stats(x, s, ss, w, b, residual, running_mean, running_var,
                                                       count, eps, momentum, relu, replicas, dropout, out,
                                                       mask_out=bits)
            if dropout is not None:
                assert relu and residual is None
                mask_mode = 3                # (the unfused backward recomputes the keep decisions from the seed)
                if out is not None:
                    holder = None
            ctx.save_for_backward(x, y if mask_mode == 1 else None, w, b, mean, invstd)
            ctx.mask_mode = mask_mode
            if holder is not None and relu:
                # what the consumer convolution's data-gradient kernel needs to run this layer's
                # ReLU mask and backward reductions in its store loop (csrc/conv_igemm2.hip, bnx)
                # (never a reference to y itself: y carries this dict as an attribute, a y -> dict -> y
                # cycle keeps every residual activation alive until Python's cyclic GC gets to it —
                # measured: peak HBM 232 GiB instead of 28 GiB at B64)
                # behind a fused dropout the mask source is the OUTPUT (dropped elements are zeros there: "a > 0"
                # is "positive and kept"), and the kept elements' gradient carries the dropout scale
                holder.update(x=x, mask_from_a=(mask_mode in (1, 3)), mean=mean, invstd=invstd, w=w, b=b,
                              mask_bits=bits)
                if dropout is not None:
                    holder["mask_scale"] = 65536.0 / (65536.0 - float(dropout[0]))
        else:
            mean = s / count
            var = (ss / count - mean * mean).clamp_min_(0.0)
            invstd = torch.rsqrt(var + eps)
            if running_mean is not None:
                with torch.no_grad():
                    unbiased = var * (count / max(count - 1.0, 1.0))
                    running_mean.mul_(1 - momentum).add_(mean.to(running_mean.dtype), alpha=momentum)
                    running_var.mul_(1 - momentum).add_(unbiased.to(running_var.dtype), alpha=momentum)
            w = _f(weight.detach()) if weight is not None else torch.ones_like(mean)
            b = _f(bias.detach()) if bias is not None else torch.zeros_like(mean)
            scale = w * invstd
            shift = b - mean * scale
            y = _t_apply(x, scale, shift, residual, relu)
            ctx.save_for_backward(x, y if relu else None, w, b, mean, invstd)
        ctx.holder = holder if (hip and holder) else None
        ctx.dropout = dropout if hip else None
        ctx.hip, ctx.relu, ctx.count, ctx.sync, ctx.pg = hip, relu, count, sync, pg
        ctx.has_res = residual is not None
        ctx.has_affine = weight is not None
        ctx.wdtype = weight.dtype if weight is not None else None
        ctx.params = (weight, bias)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, y, w, b, mean, invstd = ctx.saved_tensors
        C = x.shape[1]
        need_dres = ctx.has_res and ctx.needs_input_grad[3]
        mask_mode = ctx.mask_mode if ctx.hip else 0
        fused_dres = None
        sum_reps = None
        if ctx.hip and need_dres and ctx.mask_mode == 0:
            fused_dres, need_dres = dy, False        # no activation: the residual's gradient IS dy (no copy)
        h = ctx.holder
        drop = ctx.dropout
        if ctx.hip and h is not None and h.get("dm_ptr") == dy.data_ptr() and h.get("sums") is not None:
            # the consumer's data-gradient kernel already applied the ReLU mask to dy and left the two
            # reductions behind: no bn_bwd_reduce pass, no mask source read, the residual gradient IS dy
            part = h["sums"]                      # fp32 [reps, 2, C] partial sums
            h["sums"] = None
            if part.shape[0] > 1 and not ctx.sync:
                sum_reps = part                   # folded inside the apply kernel (no sum kernel)
            elif part.shape[0] > 1:
                part = part.sum(0, keepdim=True)  # SyncBN: one [2, C] message
            sdy, sdyx = part[0, 0], part[0, 1]
            mask_mode = 0
            drop = None                           # mask AND dropout scale were applied by that kernel
            if need_dres:
                fused_dres, need_dres = dy, False
        elif ctx.hip:
            from . import _hip_bn
            sdy, sdyx = _hip_bn.bwd_reduce(dy, x, y, mean, invstd, w, b, ctx.mask_mode, ctx.dropout)
        else:
            sdy, sdyx = _t_bwd_reduce(dy, x, y, mean, invstd, ctx.relu)
        g_sdy, g_sdyx = sdy, sdyx
        sink_scale = 1.0
        if ctx.sync and ctx.hip:
            # one message [Σdy' | Σdy'·x̂], all-reduced in place.  The affine gradients then come from the
            # GLOBAL sums scaled by 1/world: after DDP's gradient average that is the same value as
            # averaging the local sums (Σ_r local_r / W = global / W), and no copy of the local sums is kept
            if sdy.data_ptr() + 4 * C == sdyx.data_ptr():
                packed = torch.as_strided(sdy, (2, C), (C, 1))
            else:
                packed = torch.stack([sdy, sdyx])
            _small_all_reduce(packed, ctx.pg)
            sdy, sdyx = packed[0], packed[1]
            g_sdy, g_sdyx = sdy, sdyx
            sink_scale = 1.0 / dist.get_world_size(ctx.pg)
        elif ctx.sync:
            # one message [Σdy' | Σdy'·x̂]; the LOCAL sums stay behind for the affine gradients
            packed = torch.stack([sdy, sdyx])
            _small_all_reduce(packed, ctx.pg)
            g_sdy, g_sdyx = packed[0], packed[1]
        if ctx.hip:
            wp, bp = ctx.params
            gw = getattr(wp, "_dpamd_grad_sink", None) if ctx.has_affine else None
            gb = getattr(bp, "_dpamd_grad_sink", None) if ctx.has_affine else None
            use_sink = (gw is not None and gb is not None and wp.grad is not None and bp.grad is not None
                        and wp.grad.data_ptr() == gw.data_ptr() and bp.grad.data_ptr() == gb.data_ptr()
                        and ctx.needs_input_grad[1] and ctx.needs_input_grad[2])
            if use_sink:
                # affine gradients are added into the gradient arena by the same kernel
                from . import _hip_bn
                dx, dres = _hip_bn.bwd_apply(dy, x, y, mean, invstd, w, b, g_sdy, g_sdyx, ctx.count,
                                             mask_mode, need_dres, gw, gb, sdy, sdyx, drop, sink_scale,
                                             sum_replicas=sum_reps)
                if fused_dres is not None:
                    dres = fused_dres
                for prm in (wp, bp):
                    notify = getattr(prm, "_dpamd_on_grad", None)
                    if notify is not None:
                        notify(prm)
                return dx, None, None, dres, None, None, None, None, None, None, None, None, None, None
            from . import _hip_bn
            dx, dres = _hip_bn.bwd_apply(dy, x, y, mean, invstd, w, b, g_sdy, g_sdyx, ctx.count,
                                         mask_mode, need_dres, dropout=drop, sum_replicas=sum_reps)
            if fused_dres is not None:
                dres = fused_dres
            if sum_reps is not None:
                red = sum_reps.sum(0)             # autograd-owned affine gradients below need the totals
                sdy, sdyx = red[0], red[1]
        else:
            dx, dres = _t_bwd_apply(dy, x, y, mean, invstd, w, g_sdy, g_sdyx, ctx.count, ctx.relu, need_dres)
        # affine grads are *local* sums: DDP's gradient all-reduce averages them like any other param
        # clone: the sums may live in the per-step scratch pool, autograd may keep the tensor as .grad
        dweight = sdyx.to(ctx.wdtype, copy=True) if ctx.has_affine else None
        dbias = sdy.to(ctx.wdtype, copy=True) if ctx.has_affine else None
        if ctx.has_affine and sink_scale != 1.0:
            dweight, dbias = dweight * sink_scale, dbias * sink_scale
        return dx, dweight, dbias, dres, None, None, None, None, None, None, None, None, None, None


class _BatchNormActFrozen(torch.autograd.Function):
    """Eval-statistics BN (``freeze_bn``) fused with residual/ReLU; differentiable w.r.t. x."""

    @staticmethod
    def forward(ctx, x, scale, shift, residual, relu, out=None):
        y = _apply_affine(x, scale, shift, residual, relu, out)
        ctx.save_for_backward(y if relu else None, scale)
        ctx.relu, ctx.has_res = relu, residual is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        y, scale = ctx.saved_tensors
        need_dres = ctx.has_res and ctx.needs_input_grad[3]
        if B.use_hip(dy):
            from . import _hip_bn
            # frozen statistics: dx = scale·dy' — the bwd_apply kernel with (weight=scale, invstd=1,
            # mean=0, no batch terms); the ReLU mask comes from the saved output
            ones, zeros = torch.ones_like(scale), torch.zeros_like(scale)
            ref = y if y is not None else _hip_bn.as_nhwc(dy)
            dx, dres = _hip_bn.bwd_apply(dy, ref, y, zeros, ones, scale.contiguous(), zeros, None, None, 1.0,
                                         1 if ctx.relu else 0, need_dres)
            return dx, None, None, dres, None, None
        dyf = dy.float()
        if ctx.relu:
            dyf = dyf * (y > 0)
        dx = (dyf * scale.view(1, -1, 1, 1)).to(dy.dtype)
        dres = dyf.to(dy.dtype) if (ctx.has_res and ctx.needs_input_grad[3]) else None
        return dx, None, None, dres, None, None


_drop_calls = 0


def _dropout_key(p: float):
    """(thr16, seed) of one fused-dropout call: the keep probability quantised to 1/65536 and a seed
    derived from torch's seed, the rank and a per-process call counter — a kernel ARGUMENT, no device
    RNG state is read, nothing is launched."""
    global _drop_calls
    _drop_calls += 1
    rank = dist.get_rank() if (dist.is_available() and dist.is_initialized()) else 0
    seed = (torch.initial_seed() * 0x9E3779B97F4A7C15 + rank * 0xD1B54A32D192ED03 + _drop_calls * 0x94D049BB133111EB)
    thr = min(65535, max(1, int(round(p * 65536))))
    return thr, seed & 0xFFFFFFFFFFFFFFFF


def batch_norm_act(x: torch.Tensor, weight, bias, running_mean, running_var, training: bool,
                   momentum: float = 0.1, eps: float = 1e-5, relu: bool = False,
                   residual: Optional[torch.Tensor] = None, group=False,
                   conv_stats: Optional[torch.Tensor] = None, dropout: float = 0.0, out=None) -> torch.Tensor:
    """``drop(act(BN(x) + residual))``.  ``group``: ``False`` = local statistics, ``True`` = default
    process group, or a ``ProcessGroup`` = statistics all-reduced over that group.  ``dropout``: the
    probability of an ``F.dropout`` behind the activation; on the HIP training path (ReLU, no residual)
    it runs inside the BN kernels (forward and backward), elsewhere as a separate op."""
    if B.get_backend() == "torch" and not _group_active(group):
        # stock baseline: MIOpen / ATen batch-norm followed by separate add and ReLU
        y = F.batch_norm(x, running_mean, running_var, weight, bias, training, momentum, eps)
        if residual is not None:
            y = y + residual
        y = F.relu(y) if relu else y
        return F.dropout(y, dropout, training) if dropout > 0.0 else y
    if training:
        holder = {} if (relu and B.use_hip(x)) else None
        fuse_drop = dropout > 0.0 and relu and residual is None and B.use_hip(x) and _FUSE_DROPOUT
        y = _BatchNormActTrain.apply(x, weight, bias, residual, running_mean, running_var,
                                     momentum, eps, relu, group, conv_stats, holder,
                                     _dropout_key(dropout) if fuse_drop else None,
                                     out if B.use_hip(x) else None)
        if dropout > 0.0 and not fuse_drop:
            return F.dropout(y, dropout, True)
        if holder:
            y._dpamd_bn_holder = holder      # read by ops._hip_conv.conv2d(..., sole_consumer=True)
        return y
    if dropout > 0.0:
        raise ValueError("dropout with frozen / eval statistics: apply F.dropout on the result")
    invstd = torch.rsqrt(running_var.float() + eps)
    scale = invstd * (weight.float() if weight is not None else 1.0)
    shift = (bias.float() if bias is not None else 0.0) - running_mean.float() * scale
    if torch.is_grad_enabled() and x.requires_grad:
        # (its backward reads the saved output as the ReLU mask source, contiguously: no concat slot)
        return _BatchNormActFrozen.apply(x, scale, shift, residual, relu, None)
    return _apply_affine(x, scale, shift, residual, relu, out)


class BatchNorm2d(nn.Module):
    """``nn.BatchNorm2d``-compatible parameters/buffers; forward takes fusion flags.

    ``forward(x, relu=False, residual=None)`` computes ``act(BN(x) + residual)`` in one kernel pass.
    ``sync`` turns on cross-rank statistics (SyncBatchNorm); ``frozen`` uses running statistics
    even in training mode (the reference's ``freeze_bn``)."""

    def __init__(self, num_features: int, eps: float = 1e-5, momentum: float = 0.1,
                 sync: bool = False, frozen: bool = False, process_group=None):
        super().__init__()
        self.num_features, self.eps, self.momentum = num_features, eps, momentum
        self.sync, self.frozen, self.process_group = sync, frozen, process_group
        self.weight = nn.Parameter(torch.ones(num_features))
        self.bias = nn.Parameter(torch.zeros(num_features))
        self.register_buffer("running_mean", torch.zeros(num_features))
        self.register_buffer("running_var", torch.ones(num_features))
        self.register_buffer("num_batches_tracked", torch.tensor(0, dtype=torch.long))
        self._affine: Optional[torch.Tensor] = None
        self._affine_key = None

    # ---- evaluation: the running statistics as one cached (scale | shift) block ----------------------------------
    # An evaluation forward used to derive scale and shift from the running statistics on every call: five tiny ATen
    # launches per layer, 565 of the 1,400 launches of a DeepLabv3+ R101 forward and 13 % of its device time at batch 32
    # (profiles/r3_eval_forward_fused.md).  The block is computed once per evaluation phase INTO THE SAME STORAGE (a
    # captured graph keeps reading it: parallel.graphs refreshes it before a replay, as it does the bf16 weights).
    # It is stale after anything that can change the four tensors: a switch of mode (training steps update them
    # through raw pointers, which no version counter sees), a loaded state dict, a move to another device; plain
    # in-place tensor operations are caught by the version counters.
    def train(self, mode: bool = True):
        self._affine_key = None
        return super().train(mode)

    def _load_from_state_dict(self, *args, **kw):
        self._affine_key = None
        return super()._load_from_state_dict(*args, **kw)

    def _apply(self, fn, *args, **kw):
        self._affine, self._affine_key = None, None
        return super()._apply(fn, *args, **kw)

    @torch.no_grad()
    def eval_affine(self) -> torch.Tensor:
        """fp32 [2, C]: ``scale = weight / sqrt(running_var + eps)`` | ``shift = bias - running_mean * scale``."""
        key = (self.weight._version, self.bias._version, self.running_mean._version, self.running_var._version)
        a = self._affine
        if a is None or a.device != self.running_var.device:
            a = self._affine = torch.empty(2, self.num_features, dtype=torch.float32, device=self.running_var.device)
            self._affine_key = None
        if self._affine_key != key:
            invstd = torch.rsqrt(self.running_var.float() + self.eps)
            torch.mul(invstd, self.weight.float(), out=a[0])
            a[1].copy_(self.bias.float() - self.running_mean.float() * a[0])
            self._affine_key = key
        return a

    def eval_refresh(self):
        self.eval_affine()

    def forward(self, x: torch.Tensor, relu: bool = False, residual: Optional[torch.Tensor] = None,
                stats: Optional[torch.Tensor] = None, dropout: float = 0.0, out=None):
        """``dropout``: probability of a dropout behind the activation, applied in training mode only
        (fused into the BN kernels on the HIP path)."""
        training = self.training and not self.frozen
        if dropout > 0.0 and not training:
            if not self.training:
                dropout = 0.0                       # eval: identity, and the result may go to `out`
            else:                                   # frozen statistics in training mode: separate dropout
                return F.dropout(self.forward(x, relu, residual, stats), dropout, True)
        if (not self.training and dropout == 0.0 and B.get_backend() != "torch"
                and not (torch.is_grad_enabled() and x.requires_grad)):
            aff = self.eval_affine()
            return _apply_affine(x, aff[0], aff[1], residual, relu, out)
        group = False
        if self.sync and training:
            group = self.process_group if self.process_group is not None else True
        return batch_norm_act(x, self.weight, self.bias, self.running_mean, self.running_var,
                              training, self.momentum, self.eps, relu, residual, group, stats, dropout, out)

    def extra_repr(self) -> str:
        return (f"{self.num_features}, eps={self.eps}, momentum={self.momentum}, "
                f"sync={self.sync}, frozen={self.frozen}")


def convert_sync_batchnorm(module: nn.Module, process_group=None) -> nn.Module:
    """Turn on cross-rank statistics for every ``BatchNorm2d`` in ``module`` (in place)."""
    for m in module.modules():
        if isinstance(m, BatchNorm2d):
            m.sync, m.process_group = True, process_group
    return module


def freeze_batchnorm(module: nn.Module) -> nn.Module:
    for m in module.modules():
        if isinstance(m, BatchNorm2d):
            m.frozen = True
            m.weight.requires_grad_(False)
            m.bias.requires_grad_(False)
    return module
