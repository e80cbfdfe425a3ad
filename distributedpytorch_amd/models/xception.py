"""Aligned Xception-65 backbone (DeepLabv3+), ``BASELINE.json`` config 5 (output-stride 8,
"heavy atrous conv").

Every separable unit is ``depthwise 3×3 (atrous) → BN → pointwise 1×1 → BN``: the depthwise part
runs the memory-bound K13 kernel, the pointwise part the MFMA implicit-GEMM kernel (SURVEY.md
§2.5).  Activations between units are fused into the preceding BN (``relu=True``) wherever the
pre-activation structure allows it.

**Physical width of the 728-channel flow.**  728 channels are 1456-byte pixel rows: rows straddle the 128-byte
cache lines, the 64-channel slabs of the depthwise kernels share a line at every seam, and the matrix-core
kernels round 728 up to 768 anyway (3 tiles of 256 outputs, 12 K-steps of 64).  On the HIP backend the flow is
therefore STORED 768 wide (``DPAMD_X65_PAD``, `mid_channels(...)`): the 40 extra channels have zero weights, so
they carry exact zeros forward (conv of zero weights, BN of an all-zero channel, ReLU, shortcut add of two zeros)
and receive exact zeros backward (every gradient into them is a product with one of those zeros; weight decay
and momentum of zero stay zero) — the network is the 728-channel network at every step, with aligned rows.
``state_dict()`` / ``load_state_dict()`` speak the logical 728-channel shapes (hooks below), so checkpoints
stay loadable by the reference's model code.  Measured: 225.3 → 213.4 ms per step at B64, output stride 8
(`profiles/r2_x65_mid768_vs_728.log`).
"""
from __future__ import annotations

import os
from typing import Optional, Tuple

import torch
import torch.nn as nn

from ..ops import conv_bn_act, BatchNorm2d, Conv2d
from ..ops.sepconv import sepconv_bn_act


class SeparableConv2d(nn.Module):
    """'same'-padded depthwise (dilated) conv + BN + pointwise conv."""

    def __init__(self, cin: int, cout: int, stride: int = 1, dilation: int = 1):
        super().__init__()
        # effective kernel 3 + 2·(d-1); symmetric pad = d keeps 'same' for odd maps (513-derived)
        self.conv1 = Conv2d(cin, cin, 3, stride=stride, padding=dilation, dilation=dilation,
                            groups=cin, bias=False)
        self.bn = BatchNorm2d(cin)
        self.pointwise = Conv2d(cin, cout, 1, bias=False)

    def depthwise_bn(self, x):
        return conv_bn_act(self.conv1, self.bn, x)

    # evaluation constants (ops.sepconv.eval_fold_constants): stale after a switch of mode, a loaded state dict, a move
    def train(self, mode: bool = True):
        self.__dict__["_eval_fold_epoch"] = self.__dict__.get("_eval_fold_epoch", 0) + 1
        return super().train(mode)

    def _load_from_state_dict(self, *args, **kw):
        self.__dict__["_eval_fold_epoch"] = self.__dict__.get("_eval_fold_epoch", 0) + 1
        return super()._load_from_state_dict(*args, **kw)

    def _apply(self, fn, *args, **kw):
        self.__dict__.pop("_eval_fold", None)
        return super()._apply(fn, *args, **kw)

    def eval_refresh(self):
        from ..ops.sepconv import eval_fold_constants
        eval_fold_constants(self)

    def forward(self, x):
        return self.pointwise(self.depthwise_bn(x))


class _Unit(nn.Module):
    """separable conv + BN (+ fused ReLU feeding the next unit)."""

    def __init__(self, cin, cout, stride, dilation, relu_after):
        super().__init__()
        self.sep = SeparableConv2d(cin, cout, stride, dilation)
        self.bn = BatchNorm2d(cout)
        self.relu_after = relu_after

    def forward(self, x, residual=None, relu=None, passthrough=False, sole_consumer=False):
        # (GPU training path: the depthwise BatchNorm is folded into the pointwise weights, ops.sepconv)
        return sepconv_bn_act(self.sep, self.bn, x, relu=self.relu_after if relu is None else relu,
                              residual=residual, passthrough=passthrough, sole_consumer=sole_consumer)


class XceptionBlock(nn.Module):
    def __init__(self, cin: int, cout: int, reps: int, stride: int = 1, dilation: int = 1,
                 start_with_relu: bool = True, grow_first: bool = True, is_last: bool = False):
        super().__init__()
        self.start_with_relu = start_with_relu
        # set by AlignedXception: every consumer of this block's output applies a ReLU first, so the block
        # emits relu(out) from its last fused BN kernel (BN + shortcut + ReLU in one pass: no add kernel,
        # no separate ReLU forward / backward over the 728-channel maps) ...
        self.activate_output = False
        # ... and the block that receives such an input does not activate it again
        self.input_activated = False
        # set by AlignedXception: nothing but this block consumes its input tensor
        self.input_sole = False
        if cout != cin or stride != 1:
            self.skip = Conv2d(cin, cout, 1, stride=stride, bias=False)
            self.skipbn = BatchNorm2d(cout)
        else:
            self.skip = None
        plan = []  # (cin, cout, stride, dilation)
        c = cin
        if grow_first:
            plan.append((c, cout, 1, dilation))
            c = cout
        for _ in range(reps - 1):
            plan.append((c, c, 1, dilation))
        if not grow_first:
            plan.append((c, cout, 1, dilation))
        if stride != 1:
            plan.append((cout, cout, stride, 1))
        elif is_last:
            plan.append((cout, cout, 1, dilation))
        units = []
        for i, (a, b, s, d) in enumerate(plan):
            units.append(_Unit(a, b, s, d, relu_after=(i + 1 < len(plan))))
        self.rep = nn.Sequential(*units)

    def forward(self, inp):
        x = torch.relu(inp) if (self.start_with_relu and not self.input_activated) else inp
        # `x` is a fused BN+ReLU output consumed by nothing but this block: its BatchNorm's backward reductions
        # ride in the first backward kernel that sees the block's whole input gradient (ops.sepconv / ops.conv)
        sole = self.input_sole and x is inp
        units = list(self.rep)
        # pre-activation quirk of the published weights: the shortcut sees the activated input
        if self.skip is not None:
            if self.skipbn.training and not self.skipbn.frozen and self.skip.can_fuse_stats(x) \
                    and self.skip.can_passthrough(x):
                # the shortcut convolution first, the units on its passthrough: the units' input gradient is added in
                # the shortcut's data-gradient store loop (no add kernel)
                sk, st, x = self.skip(x, with_stats=True, passthrough=True, sole_consumer=sole)
                for i, unit in enumerate(units):
                    x = unit(x, sole_consumer=i > 0)
                return self.skipbn(sk, relu=self.activate_output, residual=x, stats=st)
            short = x
            for i, unit in enumerate(units):
                x = unit(x, sole_consumer=i > 0)
            return conv_bn_act(self.skip, self.skipbn, short, residual=x, relu=self.activate_output)
        x, short = units[0](x, passthrough=True, sole_consumer=sole)
        for unit in units[1:-1]:
            x = unit(x, sole_consumer=True)
        return units[-1](x, residual=short, relu=self.activate_output, sole_consumer=True)


MID_CHANNELS = 728                      # the published architecture's middle-flow width
_PADDED = ("block3.", "middle.", "block20.")   # sub-modules that hold a 728-wide dimension


def mid_channels(pad: Optional[bool] = None) -> int:
    """Physical width of the 728-channel flow: 768 (whole cache lines per pixel row, whole 64-channel slabs) when
    padding is on.  ``pad=None``: ``DPAMD_X65_PAD`` (``1``/``0``), default: on where the HIP kernels run (a GPU is present and the backend is not 'torch')."""
    if pad is None:
        env = os.environ.get("DPAMD_X65_PAD", "auto")
        if env == "auto":
            from .. import ops
            pad = ops.get_backend() != "torch" and torch.cuda.is_available()
        else:
            pad = env not in ("0", "off", "false")
    return 768 if pad else MID_CHANNELS


def _logical_state_dict(module, state_dict, prefix, local_metadata):
    """state_dict hook: the stored 768-wide tensors → the architecture's 728-wide ones (views)."""
    for key in list(state_dict.keys()):
        if not (key.startswith(prefix) and key[len(prefix):].startswith(_PADDED)):
            continue
        t = state_dict[key]
        for d in range(min(t.dim(), 2)):
            if t.shape[d] == module.mid_physical:
                t = t.narrow(d, 0, MID_CHANNELS)
        state_dict[key] = t
    return state_dict


class AlignedXception(nn.Module):
    """Returns ``(features[2048], low_level[128 @ stride 4])``."""

    def __init__(self, output_stride: int = 16, in_channels: int = 3, pad_mid: Optional[bool] = None):
        super().__init__()
        rng_at_entry = torch.get_rng_state()
        if output_stride == 16:
            entry3_stride, mid_dil, exit_dil = 2, 1, (1, 2)
        elif output_stride == 8:
            entry3_stride, mid_dil, exit_dil = 1, 2, (2, 4)
        else:
            raise ValueError("output_stride must be 8 or 16")
        self.conv1 = Conv2d(in_channels, 32, 3, stride=2, padding=1, bias=False)
        self.bn1 = BatchNorm2d(32)
        self.conv2 = Conv2d(32, 64, 3, padding=1, bias=False)
        self.bn2 = BatchNorm2d(64)
        self.block1 = XceptionBlock(64, 128, 2, stride=2, start_with_relu=False)
        self.block2 = XceptionBlock(128, 256, 2, stride=2, start_with_relu=False)
        mid = self.mid_physical = mid_channels(pad_mid)
        self.block3 = XceptionBlock(256, mid, 2, stride=entry3_stride, is_last=True)
        self.middle = nn.Sequential(*[XceptionBlock(mid, mid, 3, dilation=mid_dil) for _ in range(16)])
        self.block20 = XceptionBlock(mid, 1024, 2, dilation=exit_dil[0], grow_first=False, is_last=True)
        self.conv3 = SeparableConv2d(1024, 1536, dilation=exit_dil[1])
        self.bn3 = BatchNorm2d(1536)
        self.conv4 = SeparableConv2d(1536, 1536, dilation=exit_dil[1])
        self.bn4 = BatchNorm2d(1536)
        self.conv5 = SeparableConv2d(1536, 2048, dilation=exit_dil[1])
        self.bn5 = BatchNorm2d(2048)
        self.out_channels, self.low_level_channels = 2048, 128
        self.output_stride = output_stride
        # every block output is consumed through a ReLU (the next block's pre-activation, `low`, the
        # ReLU before conv3): produce it activated
        chain = [self.block1, self.block2, self.block3, *self.middle, self.block20]
        for blk in chain:
            blk.activate_output = True
        for blk in chain[1:]:
            blk.input_activated = True
        # block1's input (conv2 + BN + ReLU) and every block output except block1's (`low` also feeds the decoder)
        # have exactly one consumer
        for blk in chain:
            blk.input_sole = blk is not self.block2
        if mid != MID_CHANNELS:
            self._register_state_dict_hook(_logical_state_dict)
            self._register_load_state_dict_pre_hook(self._physical_state_dict)
            # Initialise as the 728-channel network does: the 768-wide layers drew kaiming_normal_ with their physical
            # fan-out (2.6 % smaller std) and consumed another stretch of the generator.  Re-draw at the logical
            # shapes from the generator state this constructor was entered with — same seed, same weights as the
            # unpadded (stock / CPU) model, and the generator is left where that model's constructor leaves it.
            torch.set_rng_state(rng_at_entry)
            twin = AlignedXception(output_stride, in_channels, pad_mid=False)
            self.load_state_dict(twin.state_dict())
            del twin
            self.zero_pad_channels()

    # ---- physical 768-wide storage of the 728-channel flow (module docstring) ----
    def _padded_tensors(self):
        """(name, tensor, dims that are `mid_physical` wide) of every parameter / buffer that has such a dim."""
        for name, t in list(self.named_parameters()) + list(self.named_buffers()):
            if name.startswith(_PADDED):
                dims = [d for d in range(min(t.dim(), 2)) if t.shape[d] == self.mid_physical]
                if dims:
                    yield name, t, dims

    @torch.no_grad()
    def zero_pad_channels(self) -> None:
        """Zero every convolution weight that touches a padding channel (after ANY re-initialisation of the
        weights): that is what makes the padding channels exact zeros in both directions."""
        for name, t, dims in self._padded_tensors():
            if t.dim() == 4 or name.endswith(".bias"):      # conv weights; BatchNorm shifts (0 at init anyway)
                for d in dims:
                    t.narrow(d, MID_CHANNELS, self.mid_physical - MID_CHANNELS).zero_()

    def logical_num_parameters(self) -> int:
        n = sum(p.numel() for p in self.parameters())
        for name, t, dims in self._padded_tensors():
            if isinstance(t, nn.Parameter):
                logical = 1
                for d in range(t.dim()):
                    logical *= MID_CHANNELS if d in dims else t.shape[d]
                n -= t.numel() - logical
        return n

    def _physical_state_dict(self, state_dict, prefix, local_metadata, strict, missing, unexpected, errors):
        """load_state_dict pre-hook: 728-wide tensors of a checkpoint → the stored width (weights / shifts /
        running means padded with 0, BatchNorm scales and running variances with 1)."""
        extra = self.mid_physical - MID_CHANNELS
        for key in list(state_dict.keys()):
            if not (key.startswith(prefix) and key[len(prefix):].startswith(_PADDED)):
                continue
            t = state_dict[key]
            for d in range(min(t.dim(), 2)):
                if t.shape[d] == MID_CHANNELS:
                    shape = list(t.shape)
                    shape[d] = extra
                    one = t.dim() == 1 and (key.endswith("running_var") or key.endswith(".weight"))
                    fill = torch.ones(shape, dtype=t.dtype, device=t.device) if one \
                        else torch.zeros(shape, dtype=t.dtype, device=t.device)
                    t = torch.cat([t, fill], dim=d)
            state_dict[key] = t

    # ---- per-parameter state kept OUTSIDE the state_dict (optimizer momentum: parallel.optim.FlatSGD) ----
    def state_to_logical(self, key: str, t: torch.Tensor) -> torch.Tensor:
        """A tensor shaped like parameter ``key`` (relative to this module) in the architecture's 728-wide shape."""
        if self.mid_physical != MID_CHANNELS and key.startswith(_PADDED):
            for d in range(min(t.dim(), 2)):
                if t.shape[d] == self.mid_physical:
                    t = t.narrow(d, 0, MID_CHANNELS)
        return t

    def state_to_physical(self, key: str, t: torch.Tensor) -> torch.Tensor:
        """The inverse for additive state (momentum): the padding channels' share is zero."""
        if self.mid_physical != MID_CHANNELS and key.startswith(_PADDED):
            for d in range(min(t.dim(), 2)):
                if t.shape[d] == MID_CHANNELS:
                    shape = list(t.shape)
                    shape[d] = self.mid_physical - MID_CHANNELS
                    t = torch.cat([t, torch.zeros(shape, dtype=t.dtype, device=t.device)], dim=d)
        return t

    def forward(self, x):
        x = conv_bn_act(self.conv1, self.bn1, x, relu=True)
        x = conv_bn_act(self.conv2, self.bn2, x, relu=True)
        low = self.block1(x)              # = relu(block output): see activate_output
        x = self.block2(low)
        x = self.block3(x)
        x = self.middle(x)
        x = self.block20(x)
        x = sepconv_bn_act(self.conv3, self.bn3, x, relu=True, sole_consumer=True)
        x = sepconv_bn_act(self.conv4, self.bn4, x, relu=True, sole_consumer=True)
        x = sepconv_bn_act(self.conv5, self.bn5, x, relu=True, sole_consumer=True)
        return x, low
