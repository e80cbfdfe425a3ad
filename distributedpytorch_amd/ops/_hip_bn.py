"""ctypes bindings for csrc/bn_relu.hip.  Host-side shape/dtype/layout checks happen HERE, before
any pointer reaches a kernel (a faulting kernel can reset the whole node)."""
from __future__ import annotations

import ctypes
from ctypes import c_float, c_int, c_long, c_void_p

import torch

from . import _backend as B
from . import workspace

_sigs_done = False


def _lib():
    global _sigs_done
    lib = B.lib()
    if not _sigs_done:
        lib.dp_bn_stats.argtypes = [c_void_p, c_long, c_int, c_void_p, c_void_p, c_void_p]
        lib.dp_bn_apply_stats.argtypes = [c_void_p, c_void_p, c_void_p, c_long, c_int] + [c_void_p] * 8 + \
            [c_float, c_float, c_float, c_int, c_int, c_long, c_void_p]
        lib.dp_bn_apply_affine.argtypes = [c_void_p, c_void_p, c_void_p, c_long, c_int, c_void_p, c_void_p,
                                           c_int, c_void_p]
        lib.dp_bn_bwd_reduce.argtypes = [c_void_p, c_void_p, c_void_p, c_long, c_int] + [c_void_p] * 6 + \
            [c_int, c_void_p]
        lib.dp_bn_bwd_apply.argtypes = [c_void_p] * 5 + [c_long, c_int] + [c_void_p] * 6 + [c_float, c_int] + \
            [c_void_p] * 4 + [c_float, c_void_p]
        lib.dp_bn_set_ld.argtypes = [c_long]
        lib.dp_bn_set_mask_out.argtypes = [c_void_p]
        lib.dp_bn_set_sum_replicas.argtypes = [c_int, c_long]
        lib.dp_bn_set_dropout.argtypes = [ctypes.c_uint, ctypes.c_ulonglong]
        lib.dp_bn_fold.argtypes = [c_void_p, c_int, c_long] + [c_void_p] * 4 + [c_float] * 3 + [c_void_p] * 7 + \
            [c_float, c_int, c_int, c_void_p, c_void_p]
        lib.dp_bn_fold_bwd.argtypes = [c_void_p] * 11 + [c_float, c_float, c_int, c_int, c_int, c_void_p]
        lib.dp_bn3fold_rows.argtypes = [c_void_p] * 3 + [c_int] + [c_void_p] * 3 + [c_int, c_int, c_void_p]
        lib.dp_bn3fold_coef.argtypes = [c_void_p] * 6 + [c_float, c_float] + [c_void_p] * 5 + [c_int, c_int, c_void_p]
        lib.dp_bn3fold_comb.argtypes = [c_void_p] * 11 + [c_int, c_int, c_void_p]
        _sigs_done = True
    return lib


def _check_act(t: torch.Tensor, name: str, like: torch.Tensor | None = None):
    if t.dtype != torch.bfloat16:
        raise TypeError(f"{name}: HIP BatchNorm path needs bf16 activations, got {t.dtype}")
    if t.dim() != 4 or not t.is_contiguous(memory_format=torch.channels_last):
        raise ValueError(f"{name}: expected a channels-last [N,C,H,W] tensor, got strides {t.stride()}")
    if t.shape[1] % 8 != 0:
        raise ValueError(f"{name}: channel count {t.shape[1]} must be a multiple of 8")
    if like is not None and t.shape != like.shape:
        raise ValueError(f"{name}: shape {tuple(t.shape)} != {tuple(like.shape)}")


def _strided(t: torch.Tensor, name: str, like: torch.Tensor):
    """(tensor, ld): ``t`` as NHWC rows of ld >= C elements — kept as is when it is a channel slice of a
    wider channels-last buffer (ops.concat), made channels-last contiguous otherwise."""
    from .concat import row_stride
    if t.dtype != torch.bfloat16:
        raise TypeError(f"{name}: HIP BatchNorm path needs bf16 activations, got {t.dtype}")
    if t.shape != like.shape:
        raise ValueError(f"{name}: shape {tuple(t.shape)} != {tuple(like.shape)}")
    ld = row_stride(t)
    if ld is None or ld % 8 != 0 or t.data_ptr() % 16 != 0:
        t = as_nhwc(t)
        ld = t.shape[1]
    return t, ld


def _arm_ld(ld: int, C: int) -> None:
    if ld != C:
        B.check(_lib().dp_bn_set_ld(int(ld)), "bn_set_ld")


def _out_slice(out, x):
    """The slot view a producer writes (ops.concat.Slot) and its row stride."""
    buf, off, c = out
    n, C, h, w = x.shape
    if (c != C or buf.dtype != x.dtype or buf.device != x.device or buf.shape[0] != n
            or tuple(buf.shape[2:]) != (h, w) or not buf.is_contiguous(memory_format=torch.channels_last)
            or off % 8 != 0 or buf.shape[1] % 8 != 0 or off + c > buf.shape[1]):
        raise ValueError("out: concat slot does not match the activation")
    return buf[:, off:off + c], buf.shape[1]


def _check_vec(t: torch.Tensor | None, C: int, name: str):
    if t is None:
        return
    if t.dtype != torch.float32 or t.numel() != C or not t.is_contiguous():
        raise ValueError(f"{name}: expected contiguous fp32[{C}], got {t.dtype}{tuple(t.shape)}")


def _mc(x):
    n, c, h, w = x.shape
    return n * h * w, c


def as_nhwc(t: torch.Tensor) -> torch.Tensor:
    return t if t.is_contiguous(memory_format=torch.channels_last) else t.contiguous(memory_format=torch.channels_last)


def stats(x: torch.Tensor):
    x = as_nhwc(x)
    _check_act(x, "x")
    M, C = _mc(x)
    buf = workspace.zeros(2 * C, x.device).view(2, C)
    B.check(_lib().dp_bn_stats(B.ptr(x), M, C, B.ptr(buf[0]), B.ptr(buf[1]), B.stream_ptr()), "bn_stats")
    return buf[0], buf[1]


def apply(x, scale, shift, residual, relu, out=None):
    """Affine form (eval / frozen statistics)."""
    x = as_nhwc(x)
    _check_act(x, "x")
    M, C = _mc(x)
    scale, shift = scale.float().contiguous(), shift.float().contiguous()
    _check_vec(scale, C, "scale"); _check_vec(shift, C, "shift")
    if residual is not None:
        residual = as_nhwc(residual)
        _check_act(residual, "residual", x)
    if out is not None:
        y, ld = _out_slice(out, x)
        _arm_ld(ld, C)
    else:
        y = torch.empty_like(x)
    B.check(_lib().dp_bn_apply_affine(B.ptr(x), B.ptr(residual), B.ptr(y), M, C, B.ptr(scale), B.ptr(shift),
                                      int(relu), B.stream_ptr()), "bn_apply_affine")
    return y


def _arm_dropout(dropout) -> None:
    """``dropout`` = (thr16, seed): the NEXT bn launch of this thread applies / recomputes the fused
    dropout (csrc/bn_relu.hip, struct Drop)."""
    if dropout is not None:
        B.check(_lib().dp_bn_set_dropout(int(dropout[0]), int(dropout[1]) & 0xFFFFFFFFFFFFFFFF), "bn_set_dropout")


def apply_from_stats(x, s, ss, weight, bias, residual, running_mean, running_var, count, eps, momentum, relu,
                     replicas=None, dropout=None, out=None, mask_out=None):
    """Training form: finalises mean/invstd/running stats inside the apply kernel.
    ``replicas``: optional fp32 [R, 2, C] partial statistics from the conv epilogue (then ``s``/``ss``
    are ignored and the kernel sums the R replicas itself).  Returns (y, mean, invstd)."""
    x = as_nhwc(x)
    _check_act(x, "x")
    M, C = _mc(x)
    nrep, rep_stride = 1, 0
    if replicas is not None:
        if replicas.dtype != torch.float32 or replicas.dim() != 3 or replicas.shape[1:] != (2, C) \
                or not replicas.is_contiguous():
            raise ValueError("replicas must be contiguous fp32 [R, 2, C]")
        nrep, rep_stride = replicas.shape[0], 2 * C
        s, ss = replicas[0, 0], replicas[0, 1]
    for t, nm in ((s, "sum"), (ss, "sqsum"), (weight, "weight"), (bias, "bias"),
                  (running_mean, "running_mean"), (running_var, "running_var")):
        _check_vec(t, C, nm)
    if residual is not None:
        residual = as_nhwc(residual)
        _check_act(residual, "residual", x)
    ld = C
    if out is not None:
        y, ld = _out_slice(out, x)
    else:
        y = torch.empty_like(x)
    mi = torch.empty(2, C, dtype=torch.float32, device=x.device)
    _arm_dropout(dropout)
    _arm_ld(ld, C)
    if mask_out is not None:
        # uint8 [M, C/8]: positivity bits of the output (csrc/bn_relu.hip, ApplyArgs::mask)
        if mask_out.dtype != torch.uint8 or mask_out.numel() != M * (C // 8) or not mask_out.is_contiguous() \
                or mask_out.device != x.device:
            raise ValueError("mask_out must be a contiguous uint8 tensor of M * C/8 bytes on x's device")
        B.check(_lib().dp_bn_set_mask_out(B.ptr(mask_out)), "bn_set_mask_out")
    B.check(_lib().dp_bn_apply_stats(B.ptr(x), B.ptr(residual), B.ptr(y), M, C, B.ptr(s), B.ptr(ss),
                                     B.ptr(weight), B.ptr(bias), B.ptr(mi[0]), B.ptr(mi[1]),
                                     B.ptr(running_mean), B.ptr(running_var), float(count), float(eps),
                                     float(momentum), int(relu), int(nrep), int(rep_stride), B.stream_ptr()),
            "bn_apply_stats")
    return y, mi[0], mi[1]


def bwd_reduce(dy, x, y, mean, invstd, weight, bias, mask_mode, dropout=None):
    _check_act(x, "x")
    dy, ld = _strided(dy, "dy", x)
    M, C = _mc(x)
    if mask_mode == 1:
        _check_act(y, "y", x)
    for t, nm in ((mean, "mean"), (invstd, "invstd"), (weight, "weight"), (bias, "bias")):
        _check_vec(t, C, nm)
    buf = workspace.zeros(2 * C, x.device).view(2, C)
    if (mask_mode == 3) != (dropout is not None):
        raise ValueError("mask_mode 3 <=> fused dropout")
    _arm_dropout(dropout)
    _arm_ld(ld, C)
    B.check(_lib().dp_bn_bwd_reduce(B.ptr(dy), B.ptr(x), B.ptr(y if mask_mode == 1 else None), M, C,
                                    B.ptr(mean), B.ptr(invstd), B.ptr(weight), B.ptr(bias),
                                    B.ptr(buf[0]), B.ptr(buf[1]), int(mask_mode), B.stream_ptr()), "bn_bwd_reduce")
    return buf[0], buf[1]


def bwd_apply(dy, x, y, mean, invstd, weight, bias, sdy, sdyx, count, mask_mode, need_dres,
              gw_sink=None, gb_sink=None, lsdy=None, lsdyx=None, dropout=None, sink_scale=1.0, sum_replicas=None):
    """``sum_replicas``: optional fp32 [R, 2, C] partial sums left by a data-gradient store loop
    (ops._hip_conv.dgrad_bn_fused); ``sdy``/``sdyx`` (and ``lsdy``/``lsdyx``) must then be its first
    replica — the kernel folds the R replicas itself, no sum kernel runs."""
    _check_act(x, "x")
    dy, ld = _strided(dy, "dy", x)
    M, C = _mc(x)
    if mask_mode == 1:
        _check_act(y, "y", x)
    for t, nm in ((mean, "mean"), (invstd, "invstd"), (weight, "weight"), (bias, "bias"),
                  (sdy, "sdy"), (sdyx, "sdyx")):
        _check_vec(t, C, nm)
    if gw_sink is not None:
        for t, nm in ((gw_sink, "gw_sink"), (gb_sink, "gb_sink"), (lsdy, "lsdy"), (lsdyx, "lsdyx")):
            _check_vec(t, C, nm)
    dx = torch.empty_like(x)
    dres = torch.empty_like(x) if need_dres else None
    if (mask_mode == 3) != (dropout is not None):
        raise ValueError("mask_mode 3 <=> fused dropout")
    if sum_replicas is not None:
        r = sum_replicas
        if r.dtype != torch.float32 or r.dim() != 3 or tuple(r.shape[1:]) != (2, C) or not r.is_contiguous() \
                or sdy is None or sdy.data_ptr() != r.data_ptr() or sdyx.data_ptr() != r.data_ptr() + 4 * C \
                or (lsdy is not None and (lsdy.data_ptr() != sdy.data_ptr() or lsdyx.data_ptr() != sdyx.data_ptr())):
            raise ValueError("sum_replicas: contiguous fp32 [R, 2, C] whose first replica is (sdy, sdyx)")
    _arm_dropout(dropout)
    _arm_ld(ld, C)
    if sum_replicas is not None and sum_replicas.shape[0] > 1:
        B.check(_lib().dp_bn_set_sum_replicas(int(sum_replicas.shape[0]), 2 * C), "bn_set_sum_replicas")
    B.check(_lib().dp_bn_bwd_apply(B.ptr(dy), B.ptr(x), B.ptr(y if mask_mode == 1 else None), B.ptr(dx),
                                   B.ptr(dres), M, C, B.ptr(mean), B.ptr(invstd), B.ptr(weight), B.ptr(bias),
                                   B.ptr(sdy), B.ptr(sdyx), float(count), int(mask_mode), B.ptr(gw_sink),
                                   B.ptr(gb_sink), B.ptr(lsdy), B.ptr(lsdyx), float(sink_scale), B.stream_ptr()),
            "bn_bwd_apply")
    return dx, dres


def fold_into_pointwise(stats, weight, bias, running_mean, running_var, count, eps, momentum, w_lowp,
                        next_running_mean=None, next_momentum=0.0, transposed=False):
    """BatchNorm (training statistics ``stats`` = fp32 [R, 2, Cin] partial sums over ``count`` rows) folded
    into the bf16 [Cout, Cin, 1, 1] weights of the 1x1 convolution that consumes it.  Returns
    (w_folded bf16 KRSC, vec fp32 [3, Cin] = scale | mean | rstd, bprime fp32 [Cout] = W @ shift).
    ``next_running_mean``: the FOLLOWING BatchNorm's running mean, pre-corrected for the constant ``bprime``
    that BN will not see (its own update then lands on the exact value).
    ``transposed``: also return the folded weights transposed (logical [Cin, Cout, 1, 1], what the data gradient
    multiplies with) as a 4th value — written by the same kernel."""
    cout, cin = w_lowp.shape[:2]
    if w_lowp.dtype != torch.bfloat16 or tuple(w_lowp.shape[2:]) != (1, 1) or cin % 8 or \
            not w_lowp.permute(0, 2, 3, 1).is_contiguous():
        raise ValueError("fold: weights must be bf16 [Cout, Cin, 1, 1] with Cin % 8 == 0")
    if stats.dtype != torch.float32 or stats.dim() != 3 or tuple(stats.shape[1:]) != (2, cin) or \
            not stats.is_contiguous():
        raise ValueError("fold: stats must be contiguous fp32 [R, 2, Cin]")
    for t, nm in ((weight, "weight"), (bias, "bias"), (running_mean, "running_mean"), (running_var, "running_var")):
        _check_vec(t, cin, nm)
    _check_vec(next_running_mean, cout, "next_running_mean")
    if next_running_mean is not None and not (0.0 <= next_momentum < 1.0):
        raise ValueError("fold: the following BatchNorm's momentum must be in [0, 1)")
    dev = w_lowp.device
    wf = torch.empty((cout, cin, 1, 1), dtype=torch.bfloat16, device=dev, memory_format=torch.channels_last)
    vec = torch.empty(3, cin, dtype=torch.float32, device=dev)
    bprime = torch.empty(cout, dtype=torch.float32, device=dev)
    rm2_scale = next_momentum / (1.0 - next_momentum) if next_running_mean is not None else 0.0
    wft = torch.empty((cin, cout, 1, 1), dtype=torch.bfloat16, device=dev, memory_format=torch.channels_last) \
        if transposed else None
    B.check(_lib().dp_bn_fold(B.ptr(stats), int(stats.shape[0]), 2 * cin, B.ptr(weight), B.ptr(bias),
                              B.ptr(running_mean), B.ptr(running_var), float(count), float(eps), float(momentum),
                              B.ptr(w_lowp), B.ptr(wf), B.ptr(vec[0]), B.ptr(vec[1]), B.ptr(vec[2]), B.ptr(bprime),
                              B.ptr(next_running_mean), float(rm2_scale), cout, cin, B.ptr(wft), B.stream_ptr()),
            "bn_fold")
    return (wf, vec, bprime, wft) if transposed else (wf, vec, bprime)


def fold_backward(dwf, w_lowp, vec, weight, dw_sink, dgamma_sink, count, sink_scale=1.0, reduce_ds=None):
    """Backward of ``fold_into_pointwise``: ``dwf`` fp32 [Cout, Cin, 1, 1] (KRSC) is the gradient w.r.t. the
    FOLDED weights.  Adds dW = dwf * scale into ``dw_sink`` and d(gamma) into ``dgamma_sink``; returns
    cc fp32 [2, Cin] = (c, -c * mean) with dz = W'^T dy + c * z + (-c * mean).
    ``reduce_ds``: synchronised statistics — called on ds = sum_k dW'_kj W_kj (fp32 [Cin]) between the two kernels
    to sum it over the ranks; ``count`` is then the global row count and ``sink_scale`` (1 / world) applies to
    d(gamma) only (the weight gradient is local)."""
    cout, cin = w_lowp.shape[:2]
    for t, nm in ((dwf, "dwf"), (dw_sink, "dw_sink")):
        if t.dtype != torch.float32 or tuple(t.shape) != (cout, cin, 1, 1) or not t.permute(0, 2, 3, 1).is_contiguous():
            raise ValueError(f"fold_backward: {nm} must be fp32 [Cout, Cin, 1, 1] stored KRSC")
    if vec.dtype != torch.float32 or tuple(vec.shape) != (3, cin) or not vec.is_contiguous():
        raise ValueError("fold_backward: vec must be fp32 [3, Cin]")
    _check_vec(weight, cin, "weight"); _check_vec(dgamma_sink, cin, "dgamma_sink")
    ds = workspace.zeros(cin, w_lowp.device)
    cc = torch.empty(2, cin, dtype=torch.float32, device=w_lowp.device)
    args = (B.ptr(dwf), B.ptr(w_lowp), B.ptr(vec[0]), B.ptr(weight), B.ptr(vec[2]), B.ptr(vec[1]), B.ptr(dw_sink),
            B.ptr(ds), B.ptr(cc[0]), B.ptr(cc[1]), B.ptr(dgamma_sink), float(count), float(sink_scale), cout, cin)
    if reduce_ds is None:
        B.check(_lib().dp_bn_fold_bwd(*args, 0, B.stream_ptr()), "bn_fold_bwd")
    else:
        B.check(_lib().dp_bn_fold_bwd(*args, 1, B.stream_ptr()), "bn_fold_bwd")
        reduce_ds(ds)
        B.check(_lib().dp_bn_fold_bwd(*args, 2, B.stream_ptr()), "bn_fold_bwd")
    return cc


def affine_add_(acc, x, scale, shift):
    """acc += x * scale + shift (per channel), in place on the channels-last bf16 tensor ``acc``."""
    _check_act(x, "x")
    _check_act(acc, "acc", x)
    M, C = _mc(x)
    _check_vec(scale, C, "scale"); _check_vec(shift, C, "shift")
    B.check(_lib().dp_bn_apply_affine(B.ptr(x), B.ptr(acc), B.ptr(acc), M, C, B.ptr(scale), B.ptr(shift), 0,
                                      B.stream_ptr()), "bn_apply_affine")
    return acc


# ------------------------------------------------------------------ BatchNorm backward folded into a 1x1 convolution
def bn3fold_rows(w_lowp, T, part, mean, rstd):
    """sums fp32 [2, K4] = (Σdm, Σ dm·x̂) from the weight gradient T = dmᵀ a (csrc/bn_relu.hip, bn3fold_rows)."""
    k4, k1 = w_lowp.shape[:2]
    if (w_lowp.dtype != torch.bfloat16 or T.dtype != torch.float32 or T.numel() != k4 * k1 or not T.is_contiguous()
            or part.dtype != torch.float32 or part.dim() != 3 or tuple(part.shape[1:]) != (2, k4)
            or not part.is_contiguous() or k1 % 8):
        raise ValueError("bn3fold_rows: W bf16 [K4, K1, 1, 1], T fp32 [K4, K1], part fp32 [R, 2, K4]")
    _check_vec(mean, k4, "mean"); _check_vec(rstd, k4, "rstd")
    sums = torch.empty(2, k4, dtype=torch.float32, device=T.device)
    B.check(_lib().dp_bn3fold_rows(B.ptr(w_lowp), B.ptr(T), B.ptr(part), int(part.shape[0]), B.ptr(mean), B.ptr(rstd),
                                   B.ptr(sums), k4, k1, B.stream_ptr()), "bn3fold_rows")
    return sums


def bn3fold_coef(w_lowp, sums, lsums, mean, rstd, gamma, count, sink_scale, dgamma_sink, dbeta_sink):
    """(Wf fp32 [K4, K1], KBW fp32 [K4, K1 + 8] = kb W | kc | 0, kvec fp32 [3, K4] = ka | kb | kc); adds the affine
    gradients into the sinks when given."""
    k4, k1 = w_lowp.shape[:2]
    for t in (sums, lsums):
        if t.dtype != torch.float32 or tuple(t.shape) != (2, k4) or not t.is_contiguous():
            raise ValueError("bn3fold_coef: sums fp32 [2, K4]")
    for t, nm in ((mean, "mean"), (rstd, "rstd"), (gamma, "gamma"), (dgamma_sink, "dgamma_sink"), (dbeta_sink, "dbeta_sink")):
        _check_vec(t, k4, nm)
    dev = w_lowp.device
    wf = torch.empty(k4, k1, dtype=torch.float32, device=dev)
    kbw = torch.empty(k4, k1 + 8, dtype=torch.float32, device=dev)      # kb W | kc | 0 (csrc/bn_relu.hip)
    kvec = torch.empty(3, k4, dtype=torch.float32, device=dev)
    B.check(_lib().dp_bn3fold_coef(B.ptr(w_lowp), B.ptr(sums), B.ptr(lsums), B.ptr(mean), B.ptr(rstd), B.ptr(gamma),
                                   1.0 / float(count), float(sink_scale), B.ptr(wf), B.ptr(kbw), B.ptr(kvec),
                                   B.ptr(dgamma_sink), B.ptr(dbeta_sink), k4, k1, B.stream_ptr()), "bn3fold_coef")
    return wf, kbw, kvec


def bn3fold_comb(T, WA, s, kvec, wf, G, dw_sink, wT):
    """dw_sink += ka T + kb WA + kc sᵀ; returns (Gb bf16 [K1, K1, 1, 1], c0 fp32 [K1], WpT bf16 [K1, K4, 1, 1])."""
    k4, k1 = wf.shape
    for t, shp, nm in ((T, k4 * k1, "T"), (WA, k4 * k1, "WA"), (G, k1 * (k1 + 8), "G"), (dw_sink, k4 * k1, "dw_sink")):
        if t.dtype != torch.float32 or t.numel() != shp or \
                not (t.is_contiguous() or (t.dim() == 4 and t.permute(0, 2, 3, 1).is_contiguous())):
            raise ValueError(f"bn3fold_comb: {nm} must be dense fp32 with {shp} elements")
    _check_vec(s, k1, "s")
    if wT.dtype != torch.bfloat16 or wT.numel() != k4 * k1 or k4 % 8:
        raise ValueError("bn3fold_comb: wT bf16 [K1, K4, 1, 1]")
    dev = wf.device
    gb = torch.empty((k1, k1, 1, 1), dtype=torch.bfloat16, device=dev, memory_format=torch.channels_last)
    wpt = torch.empty((k1, k4, 1, 1), dtype=torch.bfloat16, device=dev, memory_format=torch.channels_last)
    c0 = torch.empty(k1, dtype=torch.float32, device=dev)
    B.check(_lib().dp_bn3fold_comb(B.ptr(T), B.ptr(WA), B.ptr(s), B.ptr(kvec), B.ptr(wf), B.ptr(G), B.ptr(dw_sink),
                                   B.ptr(gb), B.ptr(c0), B.ptr(wT), B.ptr(wpt), k4, k1, B.stream_ptr()), "bn3fold_comb")
    return gb, c0, wpt
