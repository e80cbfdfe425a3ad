"""DANet's dual attention as fused operators (csrc/attention.hip).

``DANet(nclass, backbone)`` is the model the reference trains (`/root/reference/train_pascal.py:32`, `:86`);
its head (external PyTorch-Encoding package, not in the reference tree) applies a position-attention module
(self-attention over the H·W positions, q/k = C/8 channels, no 1/√d scale, ``out = γ·attn·v + x``) and a
channel-attention module (``softmax(max − X Xᵀ)`` over channels, ``out = γ·attn·X + x``).  The stock
formulation materialises an (H·W)² map per image: 64 MB fp32 + 32 MB bf16 at 64×64 positions.

GPU path: flash-style MFMA kernels (the N×N map never exists), channel attention = gram kernel + row softmax +
1×1 convolutions with per-image weights on the implicit-GEMM kernel.  CPU tensors / ``set_backend('torch')``
run the plain formulation below, which is also the reference of the numerics tests.
"""
from __future__ import annotations

import ctypes

import torch

from . import _backend as B
from ._hip_bn import as_nhwc

_sigs = False


def _lib():
    global _sigs
    lib = B.lib()
    if not _sigs:
        V, I, L, F = ctypes.c_void_p, ctypes.c_int, ctypes.c_long, ctypes.c_float
        lib.dp_pam_fwd.argtypes = [V] * 8 + [I] * 4 + [V]
        lib.dp_pam_bwd.argtypes = [V] * 11 + [I] * 4 + [V]
        lib.dp_pam_delta_parts.argtypes = [L]
        lib.dp_pam_delta_parts.restype = L
        lib.dp_pam_delta.argtypes = [V, V, V, V, L, I, V]
        lib.dp_cam_gram.argtypes = [V, V, V, I, I, I, V]
        lib.dp_cam_softmax_fwd.argtypes = [V, V, V, V, L, I, V]
        lib.dp_cam_softmax_bwd.argtypes = [V, V, V, V, V, L, I, V]
        lib.dp_cam_mat.argtypes = [V, I, V, I, I, F, F, V]
        lib.dp_head_dropout.argtypes = [V] * 6 + [L, ctypes.c_uint, ctypes.c_ulonglong, I, V]
        _sigs = True
    return lib


# ------------------------------------------------------------------------------------------ references
def position_attention_reference(q, k, v, x, gamma):
    """[n,c,h,w] tensors; the formulation of models/danet.py round 1 (fp32 softmax)."""
    n, c, h, w = v.shape
    qf = q.reshape(n, -1, h * w).transpose(1, 2)                       # [n, hw, d]
    kf = k.reshape(n, -1, h * w)                                       # [n, d, hw]
    attn = torch.softmax(torch.bmm(qf.float(), kf.float()), dim=-1)    # [n, hw, hw]
    vf = v.reshape(n, c, h * w).float()
    out = torch.bmm(vf, attn.transpose(1, 2)).reshape(n, c, h, w)
    return (gamma.float() * out + x.float()).to(x.dtype)


def channel_attention_reference(x, gamma):
    n, c, h, w = x.shape
    f = x.reshape(n, c, h * w).float()
    energy = torch.bmm(f, f.transpose(1, 2))
    energy = energy.max(dim=-1, keepdim=True)[0] - energy
    attn = torch.softmax(energy, dim=-1)
    out = torch.bmm(attn, f).reshape(n, c, h, w)
    return (gamma.float() * out + x.float()).to(x.dtype)


# ------------------------------------------------------------------------------------------ position attention
def _pad_channels(t, c):
    """Zero channels appended up to ``c`` (channels-last result; autograd slices the gradient back)."""
    if t.shape[1] == c:
        return t
    return torch.nn.functional.pad(t, (0, 0, 0, 0, 0, c - t.shape[1])).contiguous(memory_format=torch.channels_last)


class _PositionAttention(torch.autograd.Function):
    @staticmethod
    def forward(ctx, q, k, v, x, gamma):
        n, c, h, w = v.shape
        N = h * w
        q, k, v, x = as_nhwc(q), as_nhwc(k), as_nhwc(v), as_nhwc(x)
        if q.shape != (n, 64, h, w) or k.shape != q.shape or x.shape != v.shape or c % 256:
            raise ValueError("position attention kernel: q/k [n,64,h,w], v/x [n,C,h,w] with C % 256 == 0")
        g32 = gamma.detach().float().contiguous()
        o = torch.empty_like(v, memory_format=torch.channels_last)
        y = torch.empty_like(v, memory_format=torch.channels_last)
        lse = torch.empty((n, N), dtype=torch.float32, device=v.device)
        B.check(_lib().dp_pam_fwd(B.ptr(q), B.ptr(k), B.ptr(v), B.ptr(x), B.ptr(o), B.ptr(y), B.ptr(lse), B.ptr(g32),
                                  n, N, 64, c, B.stream_ptr()), "pam_fwd")
        ctx.save_for_backward(q, k, v, o, lse, g32)
        return y

    @staticmethod
    def backward(ctx, dy):
        q, k, v, o, lse, g32 = ctx.saved_tensors
        n, c, h, w = v.shape
        N = h * w
        dy = as_nhwc(dy)
        delta = torch.empty((n, N), dtype=torch.float32, device=v.device)
        lib = _lib()
        # per-wave partial sums of dgamma (no atomics: fixed summation order)
        dgp = torch.zeros(int(lib.dp_pam_delta_parts(n * N)), dtype=torch.float32, device=v.device)
        B.check(lib.dp_pam_delta(B.ptr(dy), B.ptr(o), B.ptr(delta), B.ptr(dgp), n * N, c, B.stream_ptr()),
                "pam_delta")
        # dQ: every 64-key block stores its share into a slab of its own, a second kernel sums the slabs in order
        # and rounds once (was: 2.1 GB of fp32 atomics per B32 step at 1.3 TB/s; plain stores run at 6 TB/s)
        dq_part = torch.empty(((N + 63) // 64, n, N, 64), dtype=torch.float32, device=v.device)
        dq = torch.empty_like(q, memory_format=torch.channels_last)
        dk = torch.empty_like(k, memory_format=torch.channels_last)
        dv = torch.empty_like(v, memory_format=torch.channels_last)
        B.check(lib.dp_pam_bwd(B.ptr(q), B.ptr(k), B.ptr(v), B.ptr(dy), B.ptr(lse), B.ptr(delta), B.ptr(g32),
                               B.ptr(dq_part), B.ptr(dq), B.ptr(dk), B.ptr(dv), n, N, 64, c, B.stream_ptr()), "pam_bwd")
        return dq, dk, dv, dy, dgp.sum().to(g32.dtype).view(-1)


def position_attention(q, k, v, x, gamma):
    """``γ · softmax(qᵀk) v + x`` over the H·W positions; all tensors [n, ·, h, w]."""
    if not (B.use_hip(v) and v.dtype == torch.bfloat16):
        return position_attention_reference(q, k, v, x, gamma)       # CPU tensors / the stock-op backend / fp32
    # The kernels take 64 q/k channels and 512 value channels (DANet on ResNet-50/101).
    # Other widths run the SAME kernels on zero-padded operands — exact: zero q/k channels add nothing to qᵀk, zero
    # value channels give zero outputs, which are sliced off again.  There is no quiet switch to the (H·W)² map.
    d, c = q.shape[1], v.shape[1]
    if d > 64 or k.shape[1] != d:
        raise NotImplementedError(f"position attention kernels: q/k with at most 64 channels (got {d}, {k.shape[1]}); "
                                  "set_backend('torch') runs the plain formulation")
    cp = (c + 511) // 512 * 512                    # the backward kernel keeps whole 512-wide value rows resident
    q, k, v, x = _pad_channels(q, 64), _pad_channels(k, 64), _pad_channels(v, cp), _pad_channels(x, cp)
    if cp == 512:
        y = _PositionAttention.apply(q, k, v, x, gamma)
    else:                                          # wider values: 512 channels at a time (autograd sums dq, dk)
        cl = torch.channels_last
        y = torch.cat([_PositionAttention.apply(q, k, v[:, i:i + 512].contiguous(memory_format=cl),
                                                x[:, i:i + 512].contiguous(memory_format=cl), gamma)
                       for i in range(0, cp, 512)], dim=1)
    return y if cp == c else y[:, :c]


# ------------------------------------------------------------------------------------------ channel attention
def _per_image_1x1(x, w, residual, out):
    """out[b] = x[b] · w[b]ᵀ (+ residual[b]) for every image: the implicit-GEMM conv kernel as a 1×1 convolution
    whose weights differ per image (w: [n, Cout, Cin] bf16) — one launch for the whole batch; image by image only
    where a tile would straddle two images (positions per image not a multiple of 128)."""
    from . import _hip_conv
    n, c, h, wd = x.shape
    if (h * wd) % 128 == 0:
        return _hip_conv.igemm_per_image(x, w, residual, out)
    for b in range(n):
        _hip_conv.igemm(x[b:b + 1], w[b].view(w.shape[1], w.shape[2], 1, 1), (h, wd), 1, 1, (0, 0), (1, 1),
                        residual=None if residual is None else residual[b:b + 1], out=out[b:b + 1])
    return out


class _ChannelAttention(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, gamma):
        n, c, h, w = x.shape
        N = h * w
        x = as_nhwc(x)
        if c % 128:
            raise ValueError("channel attention kernel: C % 128 == 0")
        lib = _lib()
        g32 = gamma.detach().float().contiguous()
        e = torch.zeros((n, c, c), dtype=torch.float32, device=x.device)
        B.check(lib.dp_cam_gram(B.ptr(x), B.ptr(x), B.ptr(e), n, N, c, B.stream_ptr()), "cam_gram")
        a = torch.empty((n, c, c), dtype=torch.bfloat16, device=x.device)
        ag = torch.empty_like(a)
        B.check(lib.dp_cam_softmax_fwd(B.ptr(e), B.ptr(a), B.ptr(ag), B.ptr(g32), n * c, c, B.stream_ptr()),
                "cam_softmax_fwd")
        y = torch.empty_like(x, memory_format=torch.channels_last)
        _per_image_1x1(x, ag, x, y)                 # y = x (γA)ᵀ + x : the residual add rides in the store loop
        ctx.save_for_backward(x, a, ag, g32)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, a, ag, g32 = ctx.saved_tensors
        n, c, h, w = x.shape
        N = h * w
        dy = as_nhwc(dy)
        lib = _lib()
        graw = torch.zeros((n, c, c), dtype=torch.float32, device=x.device)
        B.check(lib.dp_cam_gram(B.ptr(dy), B.ptr(x), B.ptr(graw), n, N, c, B.stream_ptr()), "cam_gram")
        de = torch.empty_like(graw)
        dgamma = torch.empty(n * c, dtype=torch.float32, device=x.device)        # per-row partials
        B.check(lib.dp_cam_softmax_bwd(B.ptr(graw), B.ptr(a), B.ptr(g32), B.ptr(de), B.ptr(dgamma), n * c, c,
                                       B.stream_ptr()), "cam_softmax_bwd")
        dgamma = dgamma.sum()
        msym = torch.empty_like(a)                  # dE + dEᵀ: E = XᵀX enters through both factors
        B.check(lib.dp_cam_mat(B.ptr(de), 0, B.ptr(msym), n, c, 1.0, 1.0, B.stream_ptr()), "cam_mat")
        agt = torch.empty_like(a)                   # (γA)ᵀ: weights of dX₁ = dy (γA)
        B.check(lib.dp_cam_mat(B.ptr(ag), 1, B.ptr(agt), n, c, 0.0, 1.0, B.stream_ptr()), "cam_mat")
        dx1 = torch.empty_like(x, memory_format=torch.channels_last)
        _per_image_1x1(dy, agt, dy, dx1)            # dy (γA) + dy   (the skip gradient rides in the store loop)
        dx = torch.empty_like(x, memory_format=torch.channels_last)
        _per_image_1x1(x, msym, dx1, dx)            # + X (dE + dEᵀ)
        return dx, dgamma.to(g32.dtype).view(-1)


def channel_attention(x, gamma):
    """``γ · softmax(max − X Xᵀ) X + x`` over channels; x [n, C, h, w]."""
    if not (B.use_hip(x) and x.dtype == torch.bfloat16):
        return channel_attention_reference(x, gamma)                  # CPU tensors / the stock-op backend / fp32
    if x.shape[1] % 128:
        # (padding is NOT exact here: a zero channel would take part in every row's softmax)
        raise NotImplementedError(f"channel attention kernels: channels must be a multiple of 128 (got {x.shape[1]}); "
                                  "set_backend('torch') runs the plain formulation")
    return _ChannelAttention.apply(x, gamma)


# ------------------------------------------------------------------------------------------ head dropouts
class _HeadDropout(torch.autograd.Function):
    """(drop(sa), drop(sc), drop(sa + sc)) with three independent masks in one pass; masks recomputed in backward."""

    @staticmethod
    def forward(ctx, sa, sc, thr, seed):
        sa, sc = as_nhwc(sa), as_nhwc(sc)
        o = [torch.empty_like(sa, memory_format=torch.channels_last) for _ in range(3)]
        B.check(_lib().dp_head_dropout(B.ptr(sa), B.ptr(sc), None, B.ptr(o[0]), B.ptr(o[1]), B.ptr(o[2]), sa.numel(),
                                       thr, seed, 0, B.stream_ptr()), "head_dropout")
        ctx.key = (thr, seed)
        return o[0], o[1], o[2]

    @staticmethod
    def backward(ctx, g0, g1, g2):
        g = [as_nhwc(t) for t in (g0, g1, g2)]
        da, dc = torch.empty_like(g[0], memory_format=torch.channels_last), \
            torch.empty_like(g[0], memory_format=torch.channels_last)
        B.check(_lib().dp_head_dropout(B.ptr(g[0]), B.ptr(g[1]), B.ptr(g[2]), B.ptr(da), B.ptr(dc), None, da.numel(),
                                       ctx.key[0], ctx.key[1], 1, B.stream_ptr()), "head_dropout_bwd")
        return da, dc, None, None


def head_dropout(sa, sc, p: float, training: bool):
    """DANet's head feeds ``drop(sa)``, ``drop(sc)`` and ``drop(sa + sc)`` to its three classifiers
    (models/danet.py); returns them in that order."""
    if not training or p <= 0.0:
        return sa, sc, sa + sc
    if B.use_hip(sa) and sa.dtype == torch.bfloat16 and sa.shape == sc.shape and sa.numel() % 8 == 0:
        from .batchnorm import _dropout_key
        thr, seed = _dropout_key(p)
        return _HeadDropout.apply(sa, sc, thr, seed)
    F = torch.nn.functional
    return F.dropout(sa, p, True), F.dropout(sc, p, True), F.dropout(sa + sc, p, True)
