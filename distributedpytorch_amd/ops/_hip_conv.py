"""Autograd wrapper + ctypes bindings for the implicit-GEMM conv kernels (csrc/conv_igemm.hip,
csrc/conv_wgrad.hip, csrc/depthwise.hip).

forward   y  = igemm(x, w)                                   [K1]
backward  dx = igemm(dy, wT, pad' = dil·(k-1) - pad, sdiv = stride)   [K2]  (same kernel)
          dw = wgrad(dy, x)   fp32, KRSC                       [K3]

All shape / dtype / layout / alignment checks are done here on the host before a pointer reaches
a kernel.
"""
from __future__ import annotations

import contextlib
import os
from ctypes import c_float, c_int, c_long, c_void_p
from typing import Optional, Tuple

import torch

from . import _backend as B
from . import streams, workspace
from ._hip_bn import as_nhwc

_sigs_done = False


def _lib():
    global _sigs_done
    lib = B.lib()
    if not _sigs_done:
        V, I = c_void_p, c_int
        lib.dp_conv_igemm.argtypes = [V, V, V, V, V] + [I] * 18 + [V]
        lib.dp_conv_igemm2.argtypes = [V, V, V, V, V] + [I] * 18 + [V]
        lib.dp_conv_igemm2r.argtypes = [V, V, V, V, V] + [I] * 18 + [V, I, V]
        lib.dp_conv_igemm2b.argtypes = [V, V, V] + [I] * 16 + [V] * 9 + [I, c_float, V]
        lib.dp_conv_igemm2ra.argtypes = [V, V, V] + [I] * 16 + [V] * 4
        lib.dp_conv_igemm2pi.argtypes = [V, V, V] + [I] * 6 + [V, V]
        if hasattr(lib, "dp_conv_igemm2e"):          # (absent from an older library named by DPAMD_HIP_LIB in A/B runs)
            lib.dp_conv_igemm2e.argtypes = [V] * 5 + [I] * 18 + [V, I, V]
        lib.dp_weight_transpose.argtypes = [V, V, I, I, I, I, V]
        lib.dp_weight_transpose_batched.argtypes = [V, V, V, I, I, V]
        lib.dp_conv_wgrad.argtypes = [V, V, V] + [I] * 16 + [V]
        lib.dp_conv_igemm2w.argtypes = [V, V, V, V, V] + [I] * 19 + [V]
        lib.dp_conv_wgrad_win.argtypes = [V, V, V] + [I] * 12 + [V]
        lib.dp_dwconv_fwd.argtypes = [V, V, V] + [I] * 10 + [V]
        lib.dp_dwconv_bwd_data_fused.argtypes = [V] * 11 + [I] * 6 + [V]
        lib.dp_dwconv_fwd_stats.argtypes = [V, V, V, V] + [I] * 11 + [V]
        lib.dp_dwconv_bwd_data.argtypes = [V, V, V] + [I] * 10 + [V]
        lib.dp_dwconv_bwd_weight.argtypes = [V, V, V] + [I] * 11 + [V]
        lib.dp_dwconv_bwd_weight_sink.argtypes = [V, V, V] + [I] * 10 + [V]
        lib.dp_dwconv_row_blocks.argtypes = [I] * 5
        lib.dp_dwconv_row_blocks.restype = c_long
        _sigs_done = True
    return lib


def _out_size(n, k, s, p, d):
    return (n + 2 * p - d * (k - 1) - 1) // s + 1


def _round8(c: int) -> int:
    return (c + 7) // 8 * 8


# The LDS-DMA kernels address their streamed operands with 32-bit byte offsets (< 2^31 - 2^24).  A batch whose
# activation passes that is cut into image chunks, one launch each, on the SAME kernels: batch statistics, fused
# BN-backward sums and weight gradients accumulate by atomics into the same buffers, so the chunks need no combine
# step.  (Round 1 sent such layers - per-GPU batch > 224 of the headline model - to the first-generation kernel.)
_OFFSET_LIMIT = (1 << 31) - (1 << 24)
_CHUNKING = True          # module switches like this one are for tests / tools to patch, not environment knobs


def _images_per_launch(n: int, *streamed) -> int:
    per_img = max(t.numel() // n * t.element_size() for t in streamed if t is not None)
    if not _CHUNKING or n <= 1 or per_img * n < _OFFSET_LIMIT:
        return n
    return max(1, (_OFFSET_LIMIT - 1) // per_img)


def igemm(x: torch.Tensor, w_krsc: torch.Tensor, out_hw: Tuple[int, int], smul: int, sdiv: int,
          pad: Tuple[int, int], dil: Tuple[int, int], bias: Optional[torch.Tensor] = None,
          stats: Optional[torch.Tensor] = None, tile: int = 0, cfg: Optional[int] = None,
          residual: Optional[torch.Tensor] = None, out: Optional[torch.Tensor] = None,
          residual_affine: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Raw kernel call.  ``x`` [N,Cin,Hi,Wi] channels-last bf16; ``w_krsc`` logical [Cout,Cin,R,S]
    stored channels-last (memory [Cout][R][S][Cin]).  Returns [N,Cout,Ho,Wo] channels-last bf16.
    ``residual`` (same shape/layout as the result) is added in the kernel's store loop;
    ``residual_affine`` (fp32 [2, Cout]: scale | shift) makes that ``residual * scale + shift``."""
    n, cin, hi, wi = x.shape
    cout, cin_w, r, s = w_krsc.shape
    if x.dtype != torch.bfloat16 or w_krsc.dtype != torch.bfloat16:
        raise TypeError("igemm: bf16 operands required")
    if cin != cin_w or cin % 8:
        raise ValueError(f"igemm: Cin mismatch or not a multiple of 8 ({cin} vs {cin_w})")
    if not x.is_contiguous(memory_format=torch.channels_last):
        raise ValueError("igemm: x must be channels-last")
    if not (w_krsc.permute(0, 2, 3, 1).is_contiguous()):
        raise ValueError("igemm: weights must be stored [Cout][R][S][Cin]")
    ho, wo = out_hw
    if out is not None:
        # caller-provided destination (e.g. one image's slice of a batch tensor: per-image weights, ops.attention)
        if (out.dtype != torch.bfloat16 or tuple(out.shape) != (n, cout, ho, wo) or out.device != x.device
                or not out.is_contiguous(memory_format=torch.channels_last) or out.data_ptr() % 16):
            raise ValueError("igemm: out must be a 16-byte aligned channels-last bf16 tensor of the result's shape")
        y = out
    else:
        y = torch.empty((n, cout, ho, wo), dtype=torch.bfloat16, device=x.device, memory_format=torch.channels_last)
    if bias is not None and (bias.dtype != torch.float32 or bias.numel() != cout):
        raise ValueError("igemm: bias must be fp32[Cout]")
    reps = 1
    if stats is not None:
        if stats.dtype != torch.float32 or stats.dim() != 3 or tuple(stats.shape[1:]) != (2, cout) \
                or not stats.is_contiguous():
            raise ValueError("igemm: stats must be contiguous fp32 [R, 2, Cout]")
        reps = stats.shape[0]
    nper = _images_per_launch(n, x)
    if nper < n:
        for n0 in range(0, n, nper):
            n1 = min(n, n0 + nper)
            igemm(x[n0:n1], w_krsc, out_hw, smul, sdiv, pad, dil, bias, stats, tile, cfg,
                  residual[n0:n1] if residual is not None else None, y[n0:n1], residual_affine)
        return y
    args = (B.ptr(x), B.ptr(w_krsc), B.ptr(y), B.ptr(bias), B.ptr(stats), reps, n, hi, wi, cin,
            ho, wo, cout, r, s, smul, sdiv, pad[0], pad[1], dil[0], dil[1], cout)
    if cfg is None:
        cfg = _FORCED_CFG
    if cfg is None:
        key = (n, hi, wi, cin, ho, wo, cout, r, s, smul, sdiv, pad, dil, bias is not None, stats is not None)
        cfg = _TUNED.get(key)
        if cfg is None and B.deterministic():
            cfg = 1                    # no timing-dependent choice: the table's tile or the default
        elif cfg is None:
            cfg = _TUNED[key] = _agreed(lambda: _autotune(args, stats, reps, cout, x.device)) if _AUTOTUNE else 1
            _save_tuned()
    if residual is not None:
        if (residual.dtype != torch.bfloat16 or residual.shape != y.shape or stats is not None
                or not residual.is_contiguous(memory_format=torch.channels_last)):
            raise ValueError("igemm: residual must be a channels-last bf16 tensor of the output's shape (no stats)")
    if residual_affine is not None:
        ra = residual_affine
        if residual is None or ra.dtype != torch.float32 or tuple(ra.shape) != (2, cout) or not ra.is_contiguous():
            raise ValueError("igemm: residual_affine must be contiguous fp32 [2, Cout] next to a residual")
    if cfg > 0:
        if residual_affine is not None:
            st = _lib().dp_conv_igemm2ra(B.ptr(x), B.ptr(w_krsc), B.ptr(y), n, hi, wi, cin, ho, wo, cout, r, s, smul,
                                         sdiv, pad[0], pad[1], dil[0], dil[1], cfg, B.ptr(residual),
                                         B.ptr(residual_affine[0]), B.ptr(residual_affine[1]), B.stream_ptr())
        elif residual is not None:
            st = _lib().dp_conv_igemm2r(*args, cfg, B.ptr(residual), cout, B.stream_ptr())
        else:
            st = _lib().dp_conv_igemm2(*args, cfg, B.stream_ptr())
        if st not in (-2, -4):   # -2: shape not taken by the second-generation kernel (bias+stats, R*S > 32);
            B.check(st, "conv_igemm2")     # -4: tensor beyond its 32-bit offsets -> first-generation kernel
            return y
    B.check(_lib().dp_conv_igemm(*args, tile, B.stream_ptr()), "conv_igemm")
    if residual_affine is not None:
        from . import _hip_bn
        return _hip_bn.affine_add_(y, residual, residual_affine[0], residual_affine[1])
    if residual is not None:
        y += residual
    return y


def igemm_infer(x: torch.Tensor, w_krsc: torch.Tensor, out_hw: Tuple[int, int], smul: int, sdiv: int,
                pad: Tuple[int, int], dil: Tuple[int, int], affine: Optional[torch.Tensor], relu: bool,
                residual: Optional[torch.Tensor] = None) -> Optional[torch.Tensor]:
    """``act(conv(x, w) * scale + shift [+ residual])`` in ONE launch (csrc/conv_igemm2.hip, ``dp_conv_igemm2e``): the
    evaluation-mode BatchNorm (``affine``: contiguous fp32 [2, Cout] = scale | shift, or None), the identity add and
    the ReLU of a block leave with the convolution's store.  Inference only (no autograd node).  Returns None for a
    shape the second-generation kernel does not take — the caller then runs convolution and BatchNorm separately."""
    n, cin, hi, wi = x.shape
    cout, cin_w, r, s = w_krsc.shape
    ho, wo = out_hw
    if (x.dtype != torch.bfloat16 or w_krsc.dtype != torch.bfloat16 or cin != cin_w or cin % 8 or cout % 8 or r * s > 32
            or not x.is_contiguous(memory_format=torch.channels_last) or not w_krsc.permute(0, 2, 3, 1).is_contiguous()
            or x.numel() * 2 >= _OFFSET_LIMIT or w_krsc.numel() * 2 >= _OFFSET_LIMIT or n * ho * wo >= (1 << 31) - 512):
        return None
    if affine is not None and (affine.dtype != torch.float32 or tuple(affine.shape) != (2, cout)
                               or not affine.is_contiguous() or affine.device != x.device):
        raise ValueError("igemm_infer: affine must be contiguous fp32 [2, Cout] on the input's device")
    if residual is not None and (residual.dtype != torch.bfloat16 or tuple(residual.shape) != (n, cout, ho, wo)
                                 or not residual.is_contiguous(memory_format=torch.channels_last)
                                 or residual.data_ptr() % 16):
        return None
    y = torch.empty((n, cout, ho, wo), dtype=torch.bfloat16, device=x.device, memory_format=torch.channels_last)
    # a tile of the second generation (the first-generation kernel, candidate 0 of the training entries, has no such
    # epilogue): the training entry of the same problem when it names one, else measured among them
    key = (n, hi, wi, cin, ho, wo, cout, r, s, smul, sdiv, pad, dil, True, False)
    cfg = _FORCED_CFG if _FORCED_CFG is not None else _TUNED.get(key)
    if cfg is not None and cfg <= 0:
        key = key + ("infer",)
        cfg = _TUNED.get(key)
    if cfg is None:
        if B.deterministic() or not _AUTOTUNE:
            cfg = 1
        else:
            shift = affine[1] if affine is not None else torch.zeros(cout, dtype=torch.float32, device=x.device)
            args = (B.ptr(x), B.ptr(w_krsc), B.ptr(y), B.ptr(shift), B.ptr(None), 1, n, hi, wi, cin, ho, wo, cout, r, s, smul,
                    sdiv, pad[0], pad[1], dil[0], dil[1], cout)
            cfg = _TUNED[key] = _agreed(lambda: _autotune(args, None, 1, cout, x.device,
                                                          candidates=[c for c in _CANDIDATES if c > 0]))
            _save_tuned()
    if cfg <= 0:
        return None
    st = _lib().dp_conv_igemm2e(B.ptr(x), B.ptr(w_krsc), B.ptr(y), B.ptr(affine[0]) if affine is not None else None,
                                B.ptr(affine[1]) if affine is not None else None, int(bool(relu)), n, hi, wi, cin, ho, wo,
                                cout, r, s, smul, sdiv, pad[0], pad[1], dil[0], dil[1], cout, cfg, B.ptr(residual), cout,
                                B.stream_ptr())
    if st in (-2, -4):
        return None
    B.check(st, "conv_igemm2e")
    return y


def igemm_per_image(x: torch.Tensor, w: torch.Tensor, residual: Optional[torch.Tensor] = None,
                    out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """``out[b] = x[b] · w[b]ᵀ (+ residual[b])``: a 1×1 convolution whose [Cout, Cin] matrix differs per image
    (``w``: contiguous bf16 [N, Cout, Cin]), as ONE launch of the implicit-GEMM kernel — a block's rows never
    straddle two images, so it points its weight descriptor at its image's matrix.  (Was a Python loop of N
    launches: ~130 per DANet step at batch 32.)"""
    n, cin, h, wd = x.shape
    if w.dim() != 3 or w.shape[0] != n or w.shape[2] != cin or w.dtype != torch.bfloat16 or not w.is_contiguous():
        raise ValueError("igemm_per_image: w must be contiguous bf16 [N, Cout, Cin]")
    cout = w.shape[1]
    if x.dtype != torch.bfloat16 or not x.is_contiguous(memory_format=torch.channels_last) or cin % 8 or cout % 8:
        raise ValueError("igemm_per_image: x must be channels-last bf16 with Cin, Cout multiples of 8")
    px = h * wd
    if px % 128:
        raise ValueError(f"igemm_per_image: {h}x{wd} positions per image is not a multiple of the 128-row tile")
    if x.numel() * 2 >= _OFFSET_LIMIT:
        raise ValueError("igemm_per_image: activation beyond the kernel's 32-bit offsets; split the batch")
    y = out if out is not None else torch.empty((n, cout, h, wd), dtype=torch.bfloat16, device=x.device,
                                                memory_format=torch.channels_last)
    for t, name in ((y, "out"), (residual, "residual")):
        if t is not None and (t.dtype != torch.bfloat16 or tuple(t.shape) != (n, cout, h, wd) or t.data_ptr() % 16
                              or not t.is_contiguous(memory_format=torch.channels_last)):
            raise ValueError(f"igemm_per_image: {name} must be a channels-last bf16 tensor of the result's shape")

    def run(cfg):
        return _lib().dp_conv_igemm2pi(B.ptr(x), B.ptr(w), B.ptr(y), n, h, wd, cin, cout, cfg, B.ptr(residual),
                                       B.stream_ptr())
    key = ("per_image", n, h, wd, cin, cout, residual is not None)
    cfg = _TUNED.get(key)
    if cfg is None:
        if B.deterministic() or not _AUTOTUNE:
            cfg = 1
        else:
            cands = [c for c in (5, 10, 3, 4, 1, 14) if px % (256 if c in (5, 10, 3) else 128) == 0]
            cfg = _TUNED[key] = _agreed(lambda: _time_best(cands, run, x.device))
            _save_tuned()
    B.check(run(cfg), "conv_igemm2pi")
    return y


def _time_best(cands, run, device) -> int:
    """Fastest of ``cands`` for ``run(cfg) -> status`` (3 timed launches each after one untimed)."""
    best, best_t = cands[-1], float("inf")
    for c in cands:
        if run(c) != 0:
            continue
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3):
            run(c)
        e1.record()
        e1.synchronize()
        t = e0.elapsed_time(e1)
        if t < best_t:
            best, best_t = c, t
    return best


# ---- tile-configuration choice -------------------------------------------------------------------
# Which tile wins depends on how the tile COUNT quantises against 256 CUs x blocks/CU for the actual
# batch (measured: the same layer-3 conv takes 35 us at 546 tiles and 22 us at 512), so the choice is
# measured once per distinct problem: every candidate runs a few times on the real operands (the
# statistics go to a scratch buffer meanwhile), the fastest is cached for the life of the process.
_FORCED_CFG = int(os.environ["DPAMD_CONV_CFG"]) if "DPAMD_CONV_CFG" in os.environ else None
_AUTOTUNE = os.environ.get("DPAMD_CONV_AUTOTUNE", "1") != "0"
# 14-16: small-LDS tiles (3-4 resident blocks per CU) for short-K layers bound by their store loop: 16 wins 23
# and 15 five of the 69 headline problems (944-952 img/s against 928 without them on the same box).  The tiles no
# tuning run ever picked (32-wide-K 4-stage rings 11/12, 8-wave small-LDS 17/18, 2, 6) were removed in round 3;
# 13 stays for the stem (profiles/r1_conv_cfg_sweep_b128.log has the sweep).
_CANDIDATES = (1, 8, 9, 3, 10, 4, 5, 7, 14, 15, 16, 0)
_COLLECTIVE = 0          # > 0 inside collective_tuning(): every rank is known to meet the same problems in the same order


class collective_tuning:
    """Section in which ALL ranks run the same launches in the same order (a training step: same model, same per-rank
    batch) — the only place where a tile choice may cost a collective.  Anywhere else a rank may be alone with a
    problem (the master's validation figure, a validation tail batch of its own size): a broadcast there waits for
    peers that never come.  Found by the 2-rank run of the entrypoint's flag matrix: rank 0 sat in the broadcast of
    ``_agreed`` inside the master-only figure forward, rank 1 in the checkpoint barrier
    (tools/gpu_runs/gpu_round3_r4y.sh)."""

    def __enter__(self):
        global _COLLECTIVE
        _COLLECTIVE += 1

    def __exit__(self, *exc):
        global _COLLECTIVE
        _COLLECTIVE -= 1
        return False


def _shared_choices() -> bool:
    """May a choice made now go into (and be read from) the table every rank holds identically?"""
    if _COLLECTIVE > 0:
        return True
    from ..parallel import dist_env
    return dist_env.get_world_size() <= 1


class _TuneTable(dict):
    """Tile choices.  The dict itself holds what is IDENTICAL on every rank: the packaged table, a tune file, choices
    agreed inside ``collective_tuning``.  ``local`` holds what this rank measured on its own outside such a section
    (several ranks only); collective sections neither read nor write it — otherwise a rank that met a problem alone
    earlier would skip the broadcast its peers enter for the same problem later."""

    def __init__(self):
        super().__init__()
        self.local: dict = {}

    def get(self, key, default=None):
        v = dict.get(self, key)
        if v is None and not _shared_choices():
            v = self.local.get(key)
        return default if v is None else v

    def __setitem__(self, key, val):
        if _shared_choices():
            dict.__setitem__(self, key, val)
        else:
            self.local[key] = val


_TUNED = _TuneTable()
_TUNE_COUNTS = {"packaged": 0, "measured_here": 0, "timed_on_this_rank": 0, "timed_outside_collective_sections": 0}


def _agreed(measure) -> int:
    """The choice for a problem that is in no table.  One process: ``measure()``.  Several ranks inside a
    ``collective_tuning`` section: rank 0 measures and everybody runs ITS choice (one small broadcast per new
    problem) — per-rank timing noise flips near-ties (20 of 101 problems between two runs of one box), and under data
    parallelism the slowest rank's pick gates every step.  Several ranks outside such a section: this rank measures
    for itself, no collective."""
    from ..parallel import dist_env
    _TUNE_COUNTS["measured_here"] += 1
    if dist_env.get_world_size() <= 1:
        _TUNE_COUNTS["timed_on_this_rank"] += 1
        return int(measure())
    if _COLLECTIVE <= 0:
        _TUNE_COUNTS["timed_on_this_rank"] += 1
        _TUNE_COUNTS["timed_outside_collective_sections"] += 1
        return int(measure())
    val = 0
    if dist_env.is_master():
        _TUNE_COUNTS["timed_on_this_rank"] += 1
        val = int(measure())
    return dist_env.broadcast_int(val, 0)


def _load_tuned(path: str) -> None:
    import json as _json
    with open(path) as _f:
        _TUNED.update({tuple(tuple(v) if isinstance(v, list) else v for v in _json.loads(k)): c
                       for k, c in _json.load(_f).items()})


# Packaged table for the headline problem (DeepLabv3+ R101, 513², per-GPU batch 128 on MI355X): the
# majority choice of three autotune runs (tools/merge_tune.py; gpurun_out/s91).  Every rank then runs the SAME tiles from the
# first step — per-rank tuning noise flips near-ties (20 of 101 problems differed between runs), and under
# data parallelism the slowest rank's choice gates every step.  Problems not in the table are measured as
# before.
_TABLE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "tuned_gfx950.json")
if os.path.isfile(_TABLE):
    _load_tuned(_TABLE)
    _TUNE_COUNTS["packaged"] = len(_TUNED)
# optional cross-process cache of the measured choices (JSON): set DPAMD_CONV_TUNE_FILE
_TUNE_FILE = os.environ.get("DPAMD_CONV_TUNE_FILE")
if _TUNE_FILE and os.path.isfile(_TUNE_FILE):
    _load_tuned(_TUNE_FILE)


def _save_tuned() -> None:
    if _TUNE_FILE:
        import json
        tmp = f"{_TUNE_FILE}.{os.getpid()}.tmp"
        with open(tmp, "w") as f:
            json.dump({json.dumps(k): c for k, c in _TUNED.items()}, f)
        os.replace(tmp, _TUNE_FILE)


def tune_report() -> dict:
    """How the tile configurations of this process were decided: ``packaged`` = entries of the shipped table
    (ops/tuned_gfx950.json) at import, ``measured_here`` = problems decided in this run (0 means every problem hit
    the table), ``timed_on_this_rank`` = how many of those this rank timed itself (all of them in one process,
    all on rank 0 / none elsewhere under data parallelism: ``_agreed``)."""
    return dict(_TUNE_COUNTS, entries=len(_TUNED), autotune=_AUTOTUNE, forced_cfg=_FORCED_CFG)


def _autotune(args, stats, reps, cout, device, runner=None, candidates=None) -> int:
    lib = _lib()
    stream = B.stream_ptr()
    a = list(args)
    if stats is not None:
        scratch = torch.zeros(reps * 2 * cout, dtype=torch.float32, device=device)
        a[4] = B.ptr(scratch)

    def run(c):
        if runner is not None:
            return runner(a, c, stream)
        if c == 0:
            return lib.dp_conv_igemm(*a, 0, stream)
        return lib.dp_conv_igemm2(*a, c, stream)

    timed = []
    for c in (candidates or _CANDIDATES):
        st = run(c)                      # warm-up (also tells whether the kernel takes this shape)
        if st in (-2, -4):               # -4: a tensor beyond the kernel's 32-bit offsets
            continue
        B.check(st, "conv autotune")
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3):
            run(c)
        e1.record()
        timed.append((c, e0, e1))
    torch.cuda.synchronize()
    if not timed:
        return 0                         # no second-generation tile takes this problem
    best = min(timed, key=lambda t: t[1].elapsed_time(t[2]))
    return best[0]


def dgrad_bn_fused(dy: torch.Tensor, wt: torch.Tensor, out_hw, sdiv: int, pad, dil, residual, holder, a=None):
    """Data gradient with the previous layer's BN+ReLU backward prologue fused into the store loop
    (csrc/conv_igemm2.hip, Conv2Args::bnx): returns ``(dm, sums)`` — the ReLU-masked gradient and the
    per-channel partial sums fp32 [reps, 2, Cout] of (Σdm, Σdm·x̂) — or ``None`` when the second-generation
    kernel does not take the shape."""
    n, cin, hi, wi = dy.shape
    cout, cin_w, r, s = wt.shape
    ho, wo = out_hw
    bx = holder.get("x")              # None: mask and Σdm only (ops.bnfold: Σdm·x̂ comes out of a weight gradient)
    shape = (n, cout, ho, wo)
    if cin != cin_w or cin % 8 or cout % 8:
        return None
    if bx is not None and (tuple(bx.shape) != shape or bx.dtype != torch.bfloat16
                           or not bx.is_contiguous(memory_format=torch.channels_last)):
        return None
    if bx is None and not holder.get("mask_from_a"):
        return None
    ba = a if holder.get("mask_from_a") else None      # the activation itself = this convolution's input
    bits = holder.get("mask_bits") if holder.get("mask_from_a") else None
    if bits is not None:
        # positivity bits left by the BatchNorm apply kernel: 1/16 of the bytes of re-reading the activation
        if bits.dtype != torch.uint8 or bits.numel() != n * ho * wo * (cout // 8) or not bits.is_contiguous():
            return None
        ba = None
    elif holder.get("mask_from_a") and ba is None:
        return None
    if ba is not None and (tuple(ba.shape) != shape or not ba.is_contiguous(memory_format=torch.channels_last)):
        return None
    if residual is not None and (tuple(residual.shape) != shape or residual.dtype != torch.bfloat16
                                 or not residual.is_contiguous(memory_format=torch.channels_last)):
        return None
    for nm in ("mean", "invstd", "w", "b"):
        t = holder.get(nm)
        if t is not None and (t.dtype != torch.float32 or t.numel() != cout or not t.is_contiguous()):
            return None
    # the fused store loop is heavier than the plain one (it reads x and the mask source): its best tile is
    # measured on the fused launch itself
    key = ("dgrad_bn", n, hi, wi, cin, ho, wo, cout, r, s, sdiv, tuple(pad), tuple(dil), residual is not None,
           ba is not None or bits is not None) + (("nox",) if bx is None else ())
    cfg = _FORCED_CFG if _FORCED_CFG is not None else _TUNED.get(key)
    y = torch.empty((n, cout, ho, wo), dtype=torch.bfloat16, device=dy.device, memory_format=torch.channels_last)
    reps = stat_replicas(n * ho * wo)

    nper = _images_per_launch(n, dy)
    bits_img = (ho * wo * (cout // 8)) if bits is not None else 0
    mask_scale = float(holder.get("mask_scale", 1.0))      # 1/(1-p): a dropout fused behind the producer's ReLU

    def launch(c, sums_ptr, stream):
        st = 0
        for n0 in range(0, n, nper):         # (one launch unless the gradient passes the kernels' 32-bit offsets)
            n1 = min(n, n0 + nper)
            sl = (lambda t: t[n0:n1] if t is not None else None)
            mask = bits.view(-1)[n0 * bits_img:] if bits is not None else None
            st = _lib().dp_conv_igemm2b(B.ptr(dy[n0:n1]), B.ptr(wt), B.ptr(y[n0:n1]), n1 - n0, hi, wi, cin, ho, wo, cout,
                                        r, s, 1, sdiv, pad[0], pad[1], dil[0], dil[1], c, B.ptr(sl(residual)),
                                        B.ptr(sl(bx)), B.ptr(sl(ba)), B.ptr(mask), B.ptr(holder.get("mean")),
                                        B.ptr(holder.get("invstd")), B.ptr(holder.get("w")), B.ptr(holder.get("b")),
                                        sums_ptr, reps, mask_scale, stream)
            if st != 0:
                break
        return st
    if cfg is None and B.deterministic():
        cfg = 1
    elif cfg is None:
        if _AUTOTUNE:
            def measure():
                scratch = torch.zeros(reps * 2 * cout, dtype=torch.float32, device=dy.device)
                return _autotune((), None, 1, cout, dy.device,
                                 runner=lambda a, c, stream: -2 if c == 0 else launch(c, B.ptr(scratch), stream))
            cfg = _agreed(measure)
        else:
            cfg = 1
        _TUNED[key] = cfg
        _save_tuned()
    if cfg <= 0:
        return None
    sums = stats_buffer(reps, cout, dy.device)
    st = launch(cfg, B.ptr(sums), B.stream_ptr())
    if st == -2:
        return None
    B.check(st, "conv_igemm2b")
    return y, fold_stats(sums)           # [reps, 2, Cout] partial sums: BN backward's apply kernel folds the replicas


def weight_transpose(w_krsc: torch.Tensor) -> torch.Tensor:
    """wT[c][R-1-r][S-1-s][k] = w[k][r][s][c]; returns logical [Cin,Cout,R,S] stored channels-last."""
    cout, cin, r, s = w_krsc.shape
    wt = torch.empty((cin, cout, r, s), dtype=w_krsc.dtype, device=w_krsc.device,
                     memory_format=torch.channels_last)
    if r == 1 and s == 1:
        # 1×1: a plain 2-D transpose
        wt.view(cin, cout).copy_(w_krsc.view(cout, cin).t())
        return wt
    B.check(_lib().dp_weight_transpose(B.ptr(w_krsc), B.ptr(wt), cout, r, s, cin, B.stream_ptr()),
            "weight_transpose")
    return wt


def weight_transpose_batched(src: torch.Tensor, dst: torch.Tensor, table: torch.Tensor, nslots: int,
                             total_tiles: int) -> None:
    """Every dense conv weight of the bf16 arena ``src`` -> tap-flipped Cin<->Cout transposed copy at the
    same offsets of ``dst`` (parallel.arena).  ``table``: uint8 view of the 32-byte slot records."""
    if src.dtype != torch.bfloat16 or dst.dtype != torch.bfloat16 or src.numel() != dst.numel():
        raise TypeError("weight_transpose_batched: bf16 arenas of equal size")
    if table.dtype != torch.uint8 or table.numel() != 32 * nslots or not table.is_contiguous():
        raise ValueError("weight_transpose_batched: table must be nslots 32-byte records")
    B.check(_lib().dp_weight_transpose_batched(B.ptr(src), B.ptr(dst), B.ptr(table), nslots, total_tiles,
                                               B.stream_ptr()), "weight_transpose_batched")


def stat_replicas(m_rows: int) -> int:
    """Replica count for the fused-statistics accumulators: ~one replica per 64 M-tiles, ≤ 8.  Deterministic mode:
    one replica per M-tile of the smallest tile height (128 rows; the kernels index ``tile_m % replicas``), so no
    two blocks add into one address; ``fold_stats`` then sums the replicas in a fixed order."""
    tiles = (m_rows + 127) // 128
    if B.deterministic():
        return max(1, tiles)
    return max(1, min(8, tiles // 64))


def stats_buffer(reps: int, c: int, device) -> torch.Tensor:
    """Zeroed fp32 [reps, 2, c] accumulator (the per-step scratch pool; a fresh tensor in deterministic mode,
    whose replica counts would drain the pool)."""
    if B.deterministic():
        return torch.zeros(reps, 2, c, dtype=torch.float32, device=device)
    return workspace.zeros(reps * 2 * c, device).view(reps, 2, c)


def fold_stats(stats: Optional[torch.Tensor]) -> Optional[torch.Tensor]:
    """Deterministic mode: the per-block replicas -> ONE [1, 2, C] block by a fixed-order sum (the consumers'
    own replica loops are meant for <= 8).  Otherwise the tensor as it is."""
    if stats is not None and B.deterministic() and stats.shape[0] > 1:
        return stats.sum(0, keepdim=True)
    return stats


def dw_row_blocks(n: int, hi: int, wi: int, ho: int, wo: int) -> int:
    """Replica count under which no two blocks of a depthwise launch share a replica (deterministic mode)."""
    return max(1, int(_lib().dp_dwconv_row_blocks(n, hi, wi, ho, wo)))


# BN+ReLU backward's mask and reductions run inside the consumer's data-gradient store loop
# (dgrad_bn_fused): the separate bn_bwd_reduce pass over dy and x disappears.  Measured A/B on one box
# (gpurun_out/s80, s81): B128 905-907 vs 880-882 img/s, B64 864 vs 858, B32 809 vs 809 (the first version,
# at B32 and before the residual/fan-out gradients moved into the store loop, was 1 % slower:
# profiles/r1_bn_prologue_ab_b32.log).  _BN_PROLOGUE = False restores the separate reduction kernel.
_BN_PROLOGUE = True


def wgrad(dy: torch.Tensor, x: torch.Tensor, wshape, stride, pad, dil, out=None, splits: int = 0) -> torch.Tensor:
    """dw fp32, logical [Cout,Cin,R,S] stored channels-last (KRSC).  ``out``: accumulate into this
    tensor (the parameter-gradient arena view) instead of a fresh zeroed one."""
    cout, cin, r, s = wshape
    n, _, hi, wi = x.shape
    _, _, ho, wo = dy.shape
    if out is not None:
        if out.dtype != torch.float32 or tuple(out.shape) != (cout, cin, r, s) or \
                not out.permute(0, 2, 3, 1).is_contiguous():
            raise ValueError("wgrad: out must be fp32 [Cout,Cin,R,S] stored KRSC")
        dw = out
    else:
        dw = torch.empty((cout, cin, r, s), dtype=torch.float32, device=x.device,
                         memory_format=torch.channels_last).zero_()
    nper = _images_per_launch(n, x, dy)
    if nper < n:
        for n0 in range(0, n, nper):         # split-K over image chunks: every launch adds into the same dw
            wgrad(dy[n0:n0 + nper], x[n0:n0 + nper], wshape, stride, pad, dil, out=dw, splits=splits)
        return dw
    if B.deterministic():
        splits = 1                       # one block per output tile: its atomics meet a zero or stream-ordered sum
    elif splits == 0 and _AUTOTUNE:
        key = ("wgrad", n, hi, wi, cin, ho, wo, cout, r, s, tuple(stride), tuple(pad), tuple(dil))
        splits = _TUNED.get(key)
        if splits is None:
            splits = _TUNED[key] = _agreed(
                lambda: _autotune_wgrad(dy, x, dw, (n, hi, wi, cin, ho, wo, cout, r, s), stride, pad, dil))
            _save_tuned()
    B.check(_lib().dp_conv_wgrad(B.ptr(dy), B.ptr(x), B.ptr(dw), n, hi, wi, cin, ho, wo, cout, r, s,
                                 stride[0], stride[1], pad[0], pad[1], dil[0], dil[1], splits, 0,
                                 B.stream_ptr()),
            "conv_wgrad")
    return dw


def _autotune_wgrad(dy, x, dw, geom, stride, pad, dil, runner=None) -> int:
    """Split-K count of the weight-gradient kernel, measured once per problem.  How well
    tiles x splits fills the 512 resident block slots (2 per CU) decides the time as much as the
    atomic-combine traffic does (gpurun_out/s18_wgrad_sweep.log: layer-3 3x3 takes 60 us at 432
    blocks, 81 us at 576), so the cost model's choice (0) competes with counts that land on
    0.5 / 0.75 / 1 / 1.5 / 2 / 3 rounds of blocks."""
    n, hi, wi, cin, ho, wo, cout, r, s = geom
    lib, stream = _lib(), B.stream_ptr()
    # blocks own 256 output channels when Cout fills them (csrc/conv_wgrad.hip, G = 2), else 128
    pad256, pad128 = (cout + 255) // 256 * 256, (cout + 127) // 128 * 128
    tk = 256 if (cout > 128 and pad256 * 8 <= pad128 * 9) else 128
    tiles = ((cout + tk - 1) // tk) * ((r * s * cin + 127) // 128)
    steps_total = (n * ho * wo + 63) // 64
    cands = [0]
    for target in (256, 384, 512, 768, 1024, 1536):
        c = max(1, min(steps_total, int(round(target / tiles))))
        if c not in cands:
            cands.append(c)
    scratch = torch.zeros_like(dw)

    def run(c):
        if runner is not None:
            return runner(B.ptr(scratch), c, stream)
        return lib.dp_conv_wgrad(B.ptr(dy), B.ptr(x), B.ptr(scratch), n, hi, wi, cin, ho, wo, cout, r, s,
                                 stride[0], stride[1], pad[0], pad[1], dil[0], dil[1], c, 0, stream)

    timed = []
    for c in cands:
        B.check(run(c), "wgrad autotune")
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3):
            run(c)
        e1.record()
        timed.append((c, e0, e1))
    torch.cuda.synchronize()
    return min(timed, key=lambda t: t[1].elapsed_time(t[2]))[0]


class _ConvIGemm(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, w_lowp, stride, padding, dilation, with_stats=False, passthrough=False,
                bn_holder=None):
        """``passthrough``: additionally return ``x`` itself as the LAST output.  A residual block
        routes its skip connection through it, so the skip-path gradient arrives in THIS backward
        and is added inside the data-gradient kernel instead of by a separate add kernel."""
        cout, cin, r, s = weight.shape
        n, _, hi, wi = x.shape
        ho = _out_size(hi, r, stride[0], padding[0], dilation[0])
        wo = _out_size(wi, s, stride[1], padding[1], dilation[1])
        if stride[0] != stride[1]:
            raise NotImplementedError("anisotropic stride")
        ctx.set_materialize_grads(False)     # no zero-fill launch for the statistics output's "gradient"
        x = as_nhwc(x)
        cin_p, cout_p = _round8(cin), _round8(cout)
        w_used = w_lowp
        if cin_p != cin or cout_p != cout:
            # rare shapes (RGB stem, 21-class classifier): zero-pad channels to the 16-byte granule
            from . import _hip_misc
            w_used = torch.empty((cout_p, cin_p, r, s), dtype=w_lowp.dtype, device=x.device,
                                 memory_format=torch.channels_last).zero_()
            w_used[:cout, :cin] = w_lowp
            if cin_p != cin:
                x = _hip_misc.pad_channels(x, cin_p)
        b32 = None
        if bias is not None:
            b32 = torch.zeros(cout_p, dtype=torch.float32, device=x.device)
            b32[:cout] = bias.detach().float()
        stats = None
        if with_stats:
            if cout_p != cout:
                raise ValueError("fused statistics need Cout % 8 == 0")
            reps = stat_replicas(n * ho * wo)
            stats = stats_buffer(reps, cout, x.device)
        y = igemm(x, w_used, (ho, wo), stride[0], 1, padding, dilation, b32, stats)
        stats = fold_stats(stats)
        ctx.save_for_backward(x, w_used)
        ctx.geom = (stride, padding, dilation, (cout, cin, r, s), (hi, wi), cout_p, cin_p)
        ctx.has_bias = bias is not None
        ctx.weight_param = weight
        ctx.with_stats, ctx.passthrough = with_stats, passthrough
        # x = relu(BN(..)) of a layer whose ONLY consumer is this convolution: its backward prologue
        # runs inside this node's data-gradient kernel (see dgrad_bn_fused)
        ctx.bn_holder = bn_holder if (bn_holder and cin_p == cin and "x" in bn_holder) else None
        if cout_p != cout:
            y = y[:, :cout]
        outs = [y]
        if with_stats:
            ctx.mark_non_differentiable(stats)
            outs.append(stats)
        if passthrough:
            if cin_p != cin:
                raise ValueError("passthrough needs Cin % 8 == 0")
            outs.append(x.view_as(x))
        return outs[0] if len(outs) == 1 else tuple(outs)

    @staticmethod
    def backward(ctx, dy, *rest):
        dskip = rest[-1] if ctx.passthrough else None
        if dy is None:
            return (dskip,) + (None,) * 9
        x, w_used = ctx.saved_tensors
        stride, padding, dilation, (cout, cin, r, s), (hi, wi), cout_p, cin_p = ctx.geom
        if cout_p != cout:
            dyp = torch.empty((dy.shape[0], cout_p, dy.shape[2], dy.shape[3]), dtype=dy.dtype, device=dy.device,
                              memory_format=torch.channels_last).zero_()
            dyp[:, :cout] = dy
            dy = dyp
        dy = as_nhwc(dy)
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            wp = ctx.weight_param
            wt = getattr(wp, "_dpamd_lowpT", None)
            # the arena's transposed copy mirrors its bf16 copy (refreshed together after the optimiser
            # step); anything else (padded channels, out-of-band weight edits) transposes here
            if not (wt is not None and getattr(wp, "_dpamd_lowpT_ok", False)
                    and w_used.data_ptr() == getattr(wp, "_dpamd_lowp", w_used).data_ptr()
                    and getattr(wp, "_dpamd_lowp", None) is not None):
                wt = weight_transpose(w_used)
            pad_t = (dilation[0] * (r - 1) - padding[0], dilation[1] * (s - 1) - padding[1])
            if dskip is not None:
                dskip = as_nhwc(dskip)
            fused = None
            if ctx.bn_holder is not None and cin_p == cin:
                fused = dgrad_bn_fused(dy, wt, (hi, wi), stride[0], pad_t, dilation, dskip, ctx.bn_holder, x)
            if fused is not None:
                dx, part = fused
                ctx.bn_holder["sums"] = part
                ctx.bn_holder["dm_ptr"] = dx.data_ptr()
            else:
                dx = igemm(dy, wt, (hi, wi), 1, stride[0], pad_t, dilation, residual=dskip)
            if cin_p != cin:
                dx = dx[:, :cin]
        elif dskip is not None:
            dx = dskip
        if ctx.needs_input_grad[1]:
            wp = ctx.weight_param
            sink = getattr(wp, "_dpamd_grad_sink", None)
            if (sink is not None and cout_p == cout and cin_p == cin and wp.grad is not None
                    and wp.grad.data_ptr() == sink.data_ptr()):
                # accumulate straight into the gradient arena (the kernel adds with fp32 atomics):
                # no temporary, no zero-fill, no AccumulateGrad add; then tell the DDP bucketer
                if _SKIP_WGRAD:
                    pass                 # diagnostics only: what the weight gradients cost the critical path
                elif streams.enabled():
                    # off the critical path: runs beside the dgrad -> BN-backward chain (ops.streams)
                    aux = streams.fork(dy.device)
                    with torch.cuda.stream(aux):
                        wgrad(dy, x, (cout, cin, r, s), stride, padding, dilation, out=sink)
                    streams.keep_alive(dy, x)
                else:
                    wgrad(dy, x, (cout, cin, r, s), stride, padding, dilation, out=sink)
                notify = getattr(wp, "_dpamd_on_grad", None)
                if notify is not None:
                    notify(wp)
            else:
                dw = wgrad(dy, x, (cout_p, cin_p, r, s), stride, padding, dilation)
                if cout_p != cout or cin_p != cin:
                    dw = dw[:cout, :cin]
        if ctx.has_bias and ctx.needs_input_grad[2]:
            # per-channel sum of the bf16 gradient in one reduction kernel (fp32 accumulation) — not an fp32 copy of the
            # whole gradient plus a reduction (DANet's value convolution: 268 MB per step for 512 numbers)
            from . import _hip_bn
            db = _hip_bn.stats(dy)[0][:cout].clone()
        return dx, dw, db, None, None, None, None, None, None, None


# ---- small-Cin stem (RGB / RGB+guidance 7x7 stride 2) ---------------------------------------------
# Zero-padding 3 channels to the 8-channel granule makes the stem a K = 49*8 GEMM of which 3/8 is
# real, on the first-generation kernel (3.6 ms per step at batch 128, profiles/r1_native_v10_b128.md).
# Instead the input is packed to 4 channels with the horizontal padding physically present, and a
# filter ROW becomes one tap: the S*4 <= 32 contiguous elements to the right of a pixel are that
# pixel's "channels" (overlapping windows, pixel stride ldx = 4).  7 taps of 32, second-generation
# kernels for forward and weight gradient, fused BN statistics as everywhere else.
_DW_WGRAD_AUX = True   # depthwise weight gradients on the aux stream
_SKIP_WGRAD = False       # diagnostics (tools patch it): invalid training, timing only
_SKIP_DW_WGRAD = False    # same, depthwise weight gradients
_STEM_WINDOW = 32
_STEM = True


def stem_window_ok(x, weight, stride, padding, dilation) -> bool:
    cout, cin, r, s = weight.shape
    return (_STEM and cin <= 4 and s * 4 <= _STEM_WINDOW and r <= 32 and cout % 8 == 0 and s > 1
            and tuple(dilation) == (1, 1) and stride[0] == stride[1] and not x.requires_grad)


def _stem_pack_weights(w: torch.Tensor) -> torch.Tensor:
    """[Cout, Cin, R, S] -> bf16 [Cout, R, 1, 32]: element s*4 + c of row r = w[k, c, r, s]."""
    cout, cin, r, s = w.shape
    wp = torch.zeros((cout, r, _STEM_WINDOW // 4, 4), dtype=torch.bfloat16, device=w.device)
    wp[:, :, :s, :cin] = w.detach().permute(0, 2, 3, 1)
    return wp.view(cout, r, 1, _STEM_WINDOW)


class _StemConv(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, stride, padding, with_stats):
        cout, cin, r, s = weight.shape
        n, _, hi, wi = x.shape
        ho = _out_size(hi, r, stride[0], padding[0], 1)
        wo = _out_size(wi, s, stride[1], padding[1], 1)
        x = as_nhwc(x)
        from . import _hip_misc
        # row width: left padding + data, and room for the last output pixel's window; even (16-byte rows)
        wp_px = max(padding[1] + wi, (wo - 1) * stride[1] + _STEM_WINDOW // 4)
        wp_px += wp_px & 1
        xp = _hip_misc.stem_pack(x, wp_px, padding[1])                       # [N, Hi, Wp, 4]
        w32 = _stem_pack_weights(weight)
        y = torch.empty((n, cout, ho, wo), dtype=torch.bfloat16, device=x.device, memory_format=torch.channels_last)
        stats = None
        reps = 1
        if with_stats:
            reps = stat_replicas(n * ho * wo)
            stats = stats_buffer(reps, cout, x.device)
        args = (B.ptr(xp), B.ptr(w32), B.ptr(y), None, B.ptr(stats), reps, n, hi, wp_px, _STEM_WINDOW, ho, wo, cout,
                r, 1, stride[0], 1, padding[0], 0, 1, 1, cout)
        key = ("stem", n, hi, wp_px, ho, wo, cout, r, stride[0], padding[0], with_stats)
        cfg = _FORCED_CFG if _FORCED_CFG else _TUNED.get(key)
        if cfg is None and B.deterministic():
            cfg = 7
        elif cfg is None:
            def runner(a, c, stream):
                return -2 if c == 0 else _lib().dp_conv_igemm2w(*a, c, 4, stream)
            cfg = _TUNED[key] = _agreed(lambda: _autotune(args, stats, reps, cout, x.device, runner,
                                                          (7, 13, 6, 1, 3, 12))) if _AUTOTUNE else 7
            _save_tuned()
        B.check(_lib().dp_conv_igemm2w(*args, cfg, 4, B.stream_ptr()), "conv_igemm2w")
        ctx.save_for_backward(xp)
        ctx.geom = (stride, padding, (cout, cin, r, s), (n, hi, wp_px, ho, wo))
        ctx.weight_param = weight
        ctx.set_materialize_grads(False)
        if with_stats:
            stats = fold_stats(stats)
            ctx.mark_non_differentiable(stats)
            return y, stats
        return y

    @staticmethod
    def backward(ctx, dy, *rest):
        if dy is None or not ctx.needs_input_grad[1]:
            return None, None, None, None, None
        (xp,) = ctx.saved_tensors
        stride, padding, (cout, cin, r, s), (n, hi, wp_px, ho, wo) = ctx.geom
        dy = as_nhwc(dy)
        dwp = workspace.zeros(cout * r * _STEM_WINDOW, dy.device)

        nper = _images_per_launch(n, dy, xp)

        def run(dst, splits, stream):
            st = 0
            for n0 in range(0, n, nper):       # (one launch unless dy passes the kernel's 32-bit offsets)
                n1 = min(n, n0 + nper)
                st = _lib().dp_conv_wgrad_win(B.ptr(dy[n0:n1]), B.ptr(xp[n0:n1]), dst, n1 - n0, hi, wp_px, _STEM_WINDOW,
                                              4, ho, wo, cout, r, stride[0], padding[0], splits, stream)
                if st != 0:
                    break
            return st
        splits = 0
        if B.deterministic():
            splits = 1
        elif _AUTOTUNE:
            key = ("stem_wgrad", n, hi, wp_px, ho, wo, cout, r, stride[0], padding[0])
            splits = _TUNED.get(key)
            if splits is None:
                splits = _TUNED[key] = _agreed(
                    lambda: _autotune_wgrad(dy, xp, dwp, (n, hi, wp_px, _STEM_WINDOW, ho, wo, cout, r, 1),
                                            stride, padding, (1, 1), run))
                _save_tuned()
        B.check(run(B.ptr(dwp), splits, B.stream_ptr()), "conv_wgrad_win")
        dw = dwp.view(cout, r, _STEM_WINDOW // 4, 4)[:, :, :s, :cin].permute(0, 3, 1, 2)    # [Cout, Cin, R, S]
        wp = ctx.weight_param
        sink = getattr(wp, "_dpamd_grad_sink", None)
        if sink is not None and wp.grad is not None and wp.grad.data_ptr() == sink.data_ptr():
            sink.add_(dw)
            notify = getattr(wp, "_dpamd_on_grad", None)
            if notify is not None:
                notify(wp)
            return None, None, None, None, None
        return None, dw.contiguous(), None, None, None


class _DepthwiseConv(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, w_lowp, stride, padding, dilation, with_stats=False, passthrough=False,
                bn_holder=None):
        """``passthrough``: additionally return ``x`` itself as the LAST output (a block routes its shortcut
        through it: the skip-path gradient is then added inside the data-gradient kernel).  ``bn_holder``: ``x`` is
        relu(BN(..)) of a layer whose whole gradient leaves this node — that layer's backward prologue (mask,
        reductions) runs in the data-gradient kernel's epilogue (csrc/depthwise.hip, MODE 1)."""
        c = x.shape[1]
        n, _, hi, wi = x.shape
        if weight.shape[2:] != (3, 3) or c % 8:
            raise NotImplementedError("depthwise kernel: 3×3, C % 8 == 0")
        x = as_nhwc(x)
        ho = _out_size(hi, 3, stride[0], padding[0], dilation[0])
        wo = _out_size(wi, 3, stride[1], padding[1], dilation[1])
        # [9][C]: tap-major so 8 channels are one 16-B load.  The arena keeps that copy of every depthwise weight
        # fresh with the optimiser step's ONE transposing launch (parallel.arena); anything else transposes here.
        w9 = getattr(weight, "_dpamd_w9", None)
        if not (w9 is not None and getattr(weight, "_dpamd_lowpT_ok", False)
                and getattr(weight, "_dpamd_lowp", None) is not None
                and w_lowp.data_ptr() == weight._dpamd_lowp.data_ptr()):
            w9 = w_lowp.reshape(c, 9).t().contiguous()
        y = torch.empty((n, c, ho, wo), dtype=x.dtype, device=x.device, memory_format=torch.channels_last)
        ctx.set_materialize_grads(False)
        stats = None
        # image chunks when a tensor passes the rolling-window kernels' 32-bit offsets (as the dense wrappers do)
        nper = _images_per_launch(n, x, y)
        spans = [(n0, min(n, n0 + nper)) for n0 in range(0, n, nper)]
        if with_stats:
            # BatchNorm statistics from the kernel's own accumulators (csrc/depthwise.hip, Dw3::stats)
            reps = dw_row_blocks(nper, hi, wi, ho, wo) if B.deterministic() else stat_replicas(n * ho * wo)
            stats = stats_buffer(reps, c, x.device)
            for i, (n0, n1) in enumerate(spans):
                st = _lib().dp_dwconv_fwd_stats(B.ptr(x[n0:n1]), B.ptr(w9), B.ptr(y[n0:n1]), B.ptr(stats), reps, n1 - n0, c,
                                                hi, wi, ho, wo, stride[0], padding[0], padding[1], dilation[0],
                                                B.stream_ptr())
                if st == -2 and i == 0:
                    stats = None                 # shape not taken by the tiled kernel: separate pass below
                    break
                B.check(st, "dwconv_fwd_stats")
        if stats is None:
            for n0, n1 in spans:
                B.check(_lib().dp_dwconv_fwd(B.ptr(x[n0:n1]), B.ptr(w9), B.ptr(y[n0:n1]), n1 - n0, c, hi, wi, ho, wo,
                                             stride[0], padding[0], padding[1], dilation[0], B.stream_ptr()), "dwconv_fwd")
        ctx.save_for_backward(x, w9)
        ctx.geom = (stride, padding, dilation, (hi, wi), (ho, wo))
        ctx.weight_param = weight
        ctx.passthrough = passthrough
        # (a holder without the pre-BN tensor, ops.bnfold, is served by the dense kernels only)
        ctx.bn_holder = bn_holder if (bn_holder and bn_holder.get("x") is not None and "mask_scale" not in bn_holder) else None
        outs = [y]
        if with_stats:
            if stats is None:
                from . import _hip_bn
                s1, s2 = _hip_bn.stats(y)
                stats = torch.stack([s1, s2]).unsqueeze(0)
            stats = fold_stats(stats)
            ctx.mark_non_differentiable(stats)
            outs.append(stats)
        if passthrough:
            outs.append(x.view_as(x))
        return outs[0] if len(outs) == 1 else tuple(outs)

    @staticmethod
    def backward(ctx, dy, *rest):
        dskip = rest[-1] if ctx.passthrough else None
        if dy is None:
            return (dskip,) + (None,) * 8
        x, w9 = ctx.saved_tensors
        stride, padding, dilation, (hi, wi), (ho, wo) = ctx.geom
        n, c = x.shape[:2]
        dy = as_nhwc(dy)
        dx = dw = None
        nper = _images_per_launch(n, x, dy)
        spans = [(n0, min(n, n0 + nper)) for n0 in range(0, n, nper)]
        if ctx.needs_input_grad[0]:
            dx = torch.empty_like(x)
            if dskip is not None:
                dskip = as_nhwc(dskip)
            h = ctx.bn_holder
            fused = False
            if (h is not None or dskip is not None) and stride[0] == 1 and padding[0] == dilation[0] \
                    and padding[1] == dilation[0] and (ho, wo) == (hi, wi):
                bx = h["x"] if h is not None else None
                if bx is not None and (tuple(bx.shape) != tuple(x.shape) or bx.dtype != torch.bfloat16
                                       or not bx.is_contiguous(memory_format=torch.channels_last)):
                    h = bx = None
                if dskip is not None and (dskip.shape != x.shape or dskip.dtype != torch.bfloat16):
                    raise ValueError("depthwise passthrough: skip gradient does not match the input")
                if bx is not None or dskip is not None:
                    sums, reps = None, 1
                    if bx is not None:
                        reps = dw_row_blocks(nper, hi, wi, ho, wo) if B.deterministic() else stat_replicas(n * hi * wi)
                        sums = stats_buffer(reps, c, x.device)
                    # mask source: the activation itself (= this node's input) where a residual entered it, else
                    # recomputed from the BatchNorm input the kernel reads anyway (one tensor less to fetch)
                    ba = x if (bx is not None and h.get("mask_from_a")) else None
                    hv = (lambda k: h.get(k) if bx is not None else None)
                    sl = (lambda t, n0, n1: t[n0:n1] if t is not None else None)
                    st = 0
                    for i, (n0, n1) in enumerate(spans):
                        st = _lib().dp_dwconv_bwd_data_fused(
                            B.ptr(dy[n0:n1]), B.ptr(w9), B.ptr(dx[n0:n1]), B.ptr(sl(dskip, n0, n1)), B.ptr(sl(ba, n0, n1)),
                            B.ptr(sl(bx, n0, n1)), B.ptr(hv("mean")), B.ptr(hv("invstd")), B.ptr(hv("w")), B.ptr(hv("b")),
                            B.ptr(sums), reps, n1 - n0, c, hi, wi, dilation[0], B.stream_ptr())
                        if st == -2 and i == 0:
                            break
                        B.check(st, "dwconv_bwd_data_fused")
                    if st != -2:
                        fused = True
                        if bx is not None:
                            h["sums"] = fold_stats(sums)
                            h["dm_ptr"] = dx.data_ptr()
            if not fused:
                for n0, n1 in spans:
                    B.check(_lib().dp_dwconv_bwd_data(B.ptr(dy[n0:n1]), B.ptr(w9), B.ptr(dx[n0:n1]), n1 - n0, c, hi, wi,
                                                      ho, wo, stride[0], padding[0], padding[1], dilation[0],
                                                      B.stream_ptr()), "dwconv_bwd_data")
                if dskip is not None:
                    dx += dskip
        elif dskip is not None:
            dx = dskip
        if ctx.needs_input_grad[1] and _SKIP_DW_WGRAD:
            notify = getattr(ctx.weight_param, "_dpamd_on_grad", None)
            if notify is not None:
                notify(ctx.weight_param)
        elif ctx.needs_input_grad[1]:
            wreps = dw_row_blocks(nper, hi, wi, ho, wo) if B.deterministic() else 1
            wp = ctx.weight_param
            sink = getattr(wp, "_dpamd_grad_sink", None)
            to_sink = sink is not None and wp.grad is not None and wp.grad.data_ptr() == sink.data_ptr()
            direct = to_sink and wreps == 1 and sink.dtype == torch.float32 and sink.numel() == 9 * c
            if direct:
                dw9r = dw9 = None
            elif wreps > 1:
                dw9r = torch.zeros(wreps, 9, c, dtype=torch.float32, device=x.device)
            else:
                dw9r = workspace.zeros(9 * c, x.device).view(1, 9, c)
            if dw9r is not None:
                dw9 = dw9r[0]
            if direct:
                # straight into the arena's [C][3][3] gradient slot (channel-major strides): no scratch, no transposing
                # add — on the auxiliary stream beside the MFMA kernels of the next backward nodes when it is on
                aux = streams.fork(dy.device) if (streams.enabled() and _DW_WGRAD_AUX) else None
                with torch.cuda.stream(aux) if aux is not None else contextlib.nullcontext():
                    for n0, n1 in spans:
                        B.check(_lib().dp_dwconv_bwd_weight_sink(B.ptr(dy[n0:n1]), B.ptr(x[n0:n1]), B.ptr(sink), n1 - n0, c,
                                                                 hi, wi, ho, wo, stride[0], padding[0], padding[1],
                                                                 dilation[0], B.stream_ptr()), "dwconv_bwd_weight_sink")
                if aux is not None:
                    streams.keep_alive(dy, x)
                notify = getattr(wp, "_dpamd_on_grad", None)
                if notify is not None:
                    notify(wp)
            elif to_sink and streams.enabled() and _DW_WGRAD_AUX:
                # off the critical path, like the dense weight gradients: the (memory-bound) reduction runs on the
                # auxiliary stream beside the MFMA kernels of the next backward nodes and lands in the arena
                aux = streams.fork(dy.device)
                with torch.cuda.stream(aux):
                    for n0, n1 in spans:
                        B.check(_lib().dp_dwconv_bwd_weight(B.ptr(dy[n0:n1]), B.ptr(x[n0:n1]), B.ptr(dw9r), wreps, n1 - n0, c,
                                                            hi, wi, ho, wo, stride[0], padding[0], padding[1], dilation[0],
                                                            B.stream_ptr()), "dwconv_bwd_weight")
                    sink.view(c, 9).add_((dw9r.sum(0) if wreps > 1 else dw9).t())
                streams.keep_alive(dy, x, dw9r)     # (dw9r: a pool slice or a fresh tensor, written and read on aux)
                notify = getattr(wp, "_dpamd_on_grad", None)
                if notify is not None:
                    notify(wp)
            else:
                for n0, n1 in spans:
                    B.check(_lib().dp_dwconv_bwd_weight(B.ptr(dy[n0:n1]), B.ptr(x[n0:n1]), B.ptr(dw9r), wreps, n1 - n0, c, hi,
                                                        wi, ho, wo, stride[0], padding[0], padding[1], dilation[0],
                                                        B.stream_ptr()), "dwconv_bwd_weight")
                if wreps > 1:
                    dw9 = dw9r.sum(0)
                dw = dw9.t().reshape(c, 1, 3, 3).contiguous()
        return dx, dw, None, None, None, None, None, None, None


def conv2d(x, weight, bias, stride, padding, dilation, groups, w_lowp, with_stats=False, passthrough=False,
           sole_consumer=False):
    if x.dtype != torch.bfloat16:
        raise TypeError(f"HIP conv path needs bf16 activations, got {x.dtype}")
    if w_lowp is None:
        w_lowp = weight.detach().to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    if groups == 1:
        if bias is None and not passthrough and stem_window_ok(x, weight, stride, padding, dilation):
            return _StemConv.apply(x, weight, stride, padding, with_stats)
        holder = getattr(x, "_dpamd_bn_holder", None) if (sole_consumer and _BN_PROLOGUE) else None
        return _ConvIGemm.apply(x, weight, bias, w_lowp, stride, padding, dilation, with_stats, passthrough,
                                holder)
    if groups == x.shape[1] == weight.shape[0] and weight.shape[1] == 1:
        if bias is not None:
            raise NotImplementedError("depthwise bias")
        if dilation[0] != dilation[1] or stride[0] != stride[1]:
            raise NotImplementedError("anisotropic depthwise")
        holder = getattr(x, "_dpamd_bn_holder", None) if (sole_consumer and _BN_PROLOGUE) else None
        return _DepthwiseConv.apply(x, weight, w_lowp, stride, padding, dilation, with_stats, passthrough, holder)
    if passthrough:
        raise NotImplementedError("passthrough: dense and depthwise convolutions only")
    if with_stats:
        raise NotImplementedError("fused statistics: dense and depthwise convolutions only")
    raise NotImplementedError(f"grouped convolution with groups={groups} (only dense and depthwise)")
