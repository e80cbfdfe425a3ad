"""Flat parameter / gradient / momentum arenas.

All trainable parameters of a model are re-homed into ONE contiguous fp32 buffer (and their
gradients, momenta and the bf16 compute copies into three more buffers with the *same element
order*).  Consequences, all of them MI355X-motivated (288 GB HBM: four flat copies of a 59 M
parameter model are ~0.8 GB, irrelevant):

* gradient buckets for the RCCL all-reduce are plain slices of the gradient arena — no
  flatten/unflatten copies;
* the optimiser is one fused kernel launch per contiguous learning-rate segment instead of ~400
  tiny per-tensor launches;
* 4-D convolution weights are stored channels-last (KRSC in memory), so the bf16 arena holds
  exactly the operand layout the implicit-GEMM kernels read, refreshed by the SGD kernel itself.

Arena order is REVERSE registration order: backward produces gradients roughly last-layer-first,
so bucket 0 fills first and its all-reduce overlaps the remaining backward.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Dict, Iterable, List, Optional, Sequence

import torch
import torch.nn as nn

_ALIGN = 64  # elements; keeps every tensor 256-byte aligned in fp32 and 128-byte in bf16


@dataclass
class Slot:
    param: nn.Parameter
    name: str
    offset: int
    numel: int
    group: int
    bucket: int = -1


@dataclass
class Bucket:
    index: int
    start: int
    end: int
    slots: List[int] = field(default_factory=list)


def _view_like(flat: torch.Tensor, offset: int, p: torch.Tensor) -> torch.Tensor:
    seg = flat[offset:offset + p.numel()]
    if p.dim() == 4:
        o, i, kh, kw = p.shape
        return seg.view(o, kh, kw, i).permute(0, 3, 1, 2)  # logical OIHW, memory KRSC
    return seg.view(p.shape)


class ParamArena:
    def __init__(self, named_param_groups: Sequence[Iterable], bucket_bytes: int = 64 << 20,
                 lowp_dtype: Optional[torch.dtype] = None, tail_bytes: int = 0, shard_align: int = 1):
        """``named_param_groups``: one iterable of ``(name, Parameter)`` per optimiser group.
        ``tail_bytes``: upper bound for the LAST bucket (the one whose all-reduce nothing overlaps; 0 = no
        special tail).  ``shard_align``: every bucket's element count is a multiple of it (reduce-scatter
        shards); slots are 64-element aligned, so any world size dividing 64 costs nothing."""
        entries = []
        seen = set()
        for gi, group in enumerate(named_param_groups):
            for name, p in group:
                if p.requires_grad and id(p) not in seen:
                    seen.add(id(p))
                    entries.append((name, p, gi))
        if not entries:
            raise ValueError("no trainable parameters")
        self._reg_index = {id(p): k for k, (_, p, _) in enumerate(entries)}
        device = entries[0][1].device
        ordered = list(reversed(entries))
        self.slots: List[Slot] = []
        off = 0
        for name, p, gi in ordered:
            self.slots.append(Slot(p, name, off, p.numel(), gi))
            off += (p.numel() + _ALIGN - 1) // _ALIGN * _ALIGN
        self.total = off
        self.device = device
        self.flat_p = torch.zeros(off, dtype=torch.float32, device=device)
        self.flat_g = torch.zeros(off, dtype=torch.float32, device=device)
        self.flat_m = torch.zeros(off, dtype=torch.float32, device=device)
        self.flat_lowp = torch.zeros(off, dtype=lowp_dtype, device=device) if lowp_dtype else None
        self.lowp_views: Dict[int, torch.Tensor] = {}
        with torch.no_grad():
            for s in self.slots:
                view = _view_like(self.flat_p, s.offset, s.param)
                view.copy_(s.param.data)
                s.param.data = view
                s.param.grad = _view_like(self.flat_g, s.offset, s.param)
                # kernels that produce this parameter's gradient add straight into the arena
                s.param._dpamd_grad_sink = s.param.grad
                if self.flat_lowp is not None:
                    lv = _view_like(self.flat_lowp, s.offset, s.param)
                    self.lowp_views[id(s.param)] = lv
                    # consumed by ops.Conv2d.lowp_weight: (view, parameter version it mirrors)
                    s.param._dpamd_lowp = lv
            if self.flat_lowp is not None:
                self.flat_lowp.copy_(self.flat_p)
                for s in self.slots:
                    s.param._dpamd_lowp_version = s.param._version
        self._init_transposed()
        # ---- buckets: contiguous slot runs of ~bucket_bytes ----
        self.buckets: List[Bucket] = []
        cap = max(bucket_bytes // 4, 1)
        cur = Bucket(0, 0, 0)
        for si, s in enumerate(self.slots):
            span_end = s.offset + (s.numel + _ALIGN - 1) // _ALIGN * _ALIGN
            if cur.slots and span_end - cur.start > cap:
                self.buckets.append(cur)
                cur = Bucket(len(self.buckets), s.offset, s.offset)
            cur.slots.append(si)
            cur.end = span_end
            s.bucket = cur.index
        self.buckets.append(cur)
        # ---- small tail: split the last bucket so that its final run is at most tail_bytes ----
        tail_cap = tail_bytes // 4
        last = self.buckets[-1]
        if tail_cap > 0 and last.end - last.start > tail_cap and len(last.slots) > 1:
            k = len(last.slots) - 1          # the tail always keeps the last slot
            while k > 1 and last.end - self.slots[last.slots[k - 1]].offset <= tail_cap:
                k -= 1
            head_slots, tail_slots = last.slots[:k], last.slots[k:]
            cut = self.slots[tail_slots[0]].offset
            self.buckets[-1] = Bucket(last.index, last.start, cut, head_slots)
            tail = Bucket(len(self.buckets), cut, last.end, tail_slots)
            self.buckets.append(tail)
            for si in tail_slots:
                self.slots[si].bucket = tail.index
        if shard_align > 1 and any((b.end - b.start) % shard_align for b in self.buckets):
            if _ALIGN % shard_align:
                raise ValueError(f"world size {shard_align} does not divide the arena alignment {_ALIGN}")
        # ---- optimiser segments: maximal runs of equal group ----
        self.segments: List[tuple] = []  # (start, end, group)
        st, g = 0, self.slots[0].group
        for s in self.slots[1:]:
            if s.group != g:
                self.segments.append((st, s.offset, g))
                st, g = s.offset, s.group
        self.segments.append((st, self.total, g))

    # ------------------------------------------------------------------
    def bucket_grad(self, b: int) -> torch.Tensor:
        bk = self.buckets[b]
        return self.flat_g[bk.start:bk.end]

    def refresh_lowp(self) -> None:
        if self.flat_lowp is not None:
            with torch.no_grad():
                self.flat_lowp.copy_(self.flat_p)
            for s in self.slots:
                s.param._dpamd_lowp_version = s.param._version
        self.refresh_transposed()

    # ------------------------------------------------------------------
    # Data-gradient weights.  conv dgrad runs the forward kernel on the tap-flipped, Cin<->Cout
    # transposed weights; building that copy per layer inside backward costs a launch per convolution
    # on the critical path (113 for DeepLabv3+ R101).  The arena keeps a third bf16 buffer with the
    # same offsets and refreshes ALL of it with one kernel right after the optimiser step.
    def _init_transposed(self) -> None:
        self.flat_lowpT = None
        self._wt_table = None
        if self.flat_lowp is None or self.flat_lowp.dtype != torch.bfloat16 or not self.flat_lowp.is_cuda:
            return
        import numpy as np
        recs, tile0 = [], 0
        dense, depthwise = [], []
        for s in self.slots:
            p = s.param
            if p.dim() == 4 and p.shape[1] > 1 and p.shape[0] % 8 == 0 and p.shape[1] % 8 == 0:
                k, c, r, sw = p.shape
                tiles_k = (k + 31) // 32
                recs.append((s.offset, k, r, sw, c, tile0, tiles_k))
                tile0 += r * sw * tiles_k * ((c + 31) // 32)
                dense.append(s)
            elif p.dim() == 4 and p.shape[1] == 1 and tuple(p.shape[2:]) == (3, 3) and p.shape[0] % 8 == 0:
                # depthwise [C,1,3,3]: the kernels read tap-major [9][C] weights (8 channels = one 16-byte load); the
                # same launch keeps that copy fresh (R < 0: no tap flip) instead of one transposing copy per layer
                # and step in forward (65 launches per Xception-65 step)
                k = p.shape[0]
                tiles_k = (k + 31) // 32
                recs.append((s.offset, k, -3, 3, 1, tile0, tiles_k))
                tile0 += 9 * tiles_k
                depthwise.append(s)
        if not recs:
            return
        tab = np.zeros(len(recs), dtype=np.dtype([("off", "<i8"), ("K", "<i4"), ("R", "<i4"), ("S", "<i4"),
                                                  ("C", "<i4"), ("tile0", "<i4"), ("tiles_k", "<i4")]))
        for i, rec in enumerate(recs):
            tab[i] = rec
        self._wt_table = torch.from_numpy(tab.view(np.uint8).copy()).to(self.device)
        self._wt_n, self._wt_tiles = len(recs), tile0
        self.flat_lowpT = torch.zeros(self.total, dtype=torch.bfloat16, device=self.device)
        for s in dense:
            k, c, r, sw = s.param.shape
            seg = self.flat_lowpT[s.offset:s.offset + s.numel]
            # logical [Cin, Cout, R, S] stored channels-last = memory [c][r][s][k]
            s.param._dpamd_lowpT = seg.view(c, r, sw, k).permute(0, 3, 1, 2)
            s.param._dpamd_lowpT_ok = False
        for s in depthwise:
            s.param._dpamd_w9 = self.flat_lowpT[s.offset:s.offset + s.numel].view(9, s.param.shape[0])
            s.param._dpamd_lowpT_ok = False
        self._wt_dense = dense + depthwise
        self.refresh_transposed()

    def refresh_transposed(self) -> None:
        if getattr(self, "flat_lowpT", None) is None:
            return
        from ..ops import _backend as B
        from ..ops import _hip_conv
        _hip_conv.weight_transpose_batched(self.flat_lowp, self.flat_lowpT, self._wt_table, self._wt_n,
                                           self._wt_tiles)
        for s in self._wt_dense:
            s.param._dpamd_lowpT_ok = True

    def zero_grad(self) -> None:
        self.flat_g.zero_()

    def describe(self) -> str:
        mb = self.total * 4 / 2 ** 20
        return (f"ParamArena: {len(self.slots)} tensors, {self.total} elems ({mb:.1f} MiB fp32), "
                f"{len(self.buckets)} buckets, {len(self.segments)} optimiser segments")
