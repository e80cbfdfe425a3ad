"""Prediction on images that carry no annotation: the serving side of the two tasks.

The reference only ever runs its network inside the training script's validation loop
(`/root/reference/train_pascal.py:237-298`): batch 1, ground-truth masks to find the crop, one object at a time.
These two classes run the same chains WITHOUT ground truth, batched and on the device:

``SemSegPredictor``   images of any size -> class maps of the same size.  One gather kernel letterboxes (or
                      stretches) a padded uint8 batch to ``crop`` x ``crop`` and normalises it (ops.affine_augment),
                      the forward is replayed from a hipGraph per batch shape (parallel.graphs: the batch-1 forward
                      is launch-bound, 120 -> 261 img/s), logits go back to each image's own size bilinearly.
``InstancePredictor`` image + four extreme points of an object -> probability map of the object at the image's size:
                      box of the points -> ``relax``-50 crop -> bicubic resize -> n-ellipse guidance channel -> concat ->
                      forward -> sigmoid -> paste back (`/root/reference/train_pascal.py:135-145`, `:280-290`, with the
                      clicked points in place of the ones the reference derives from the ground-truth mask).

Both work on the CPU too (fp32, the PyTorch forms of the same ops): that is what the CPU tests run.
"""
from __future__ import annotations

from typing import List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from .ops import affine_augment
from .ops.augment import _identity_mats, affine_augment2, compose_crop, nellipse_guidance_



def voc_palette(n: int = 256) -> np.ndarray:
    """uint8 [n, 3]: the PASCAL VOC label colours (bit-reversed class index spread over the three channels)."""
    pal = np.zeros((n, 3), dtype=np.uint8)
    for i in range(n):
        c, r, g, b = i, 0, 0, 0
        for j in range(8):
            r |= ((c >> 0) & 1) << (7 - j)
            g |= ((c >> 1) & 1) << (7 - j)
            b |= ((c >> 2) & 1) << (7 - j)
            c >>= 3
        pal[i] = (r, g, b)
    return pal


def _as_u8_rgb(img) -> np.ndarray:
    a = np.asarray(img)
    if a.ndim == 2:
        a = np.stack([a] * 3, axis=-1)
    if a.ndim != 3 or a.shape[2] < 3:
        raise ValueError(f"image must be HxWx3 (got {a.shape})")
    a = a[..., :3]
    if a.dtype != np.uint8:
        a = np.clip(np.rint(a.astype(np.float32)), 0, 255).astype(np.uint8)
    return a


class _Base:
    def __init__(self, model: nn.Module, device=None, graph: bool = True):
        if device is None:
            device = torch.device("cuda", torch.cuda.current_device()) if torch.cuda.is_available() else torch.device("cpu")
        self.device = torch.device(device)
        self.on_gpu = self.device.type == "cuda"
        self.dtype = torch.bfloat16 if self.on_gpu else torch.float32
        if any(hasattr(p, "_dpamd_grad_sink") for p in model.parameters()):
            # a TrainEngine's model: its parameters are views of the engine's arena, already on the device and in the
            # kernels' layout — moving or re-laying them out would cut them loose from it
            self.model = model.eval()
        else:
            self.model = model.to(self.device).to(memory_format=torch.channels_last).eval()
        self._fwd = self.model
        if graph and self.on_gpu:
            from .parallel.graphs import GraphedForward
            self._fwd = GraphedForward(self.model)

    def _canvas(self, images: Sequence[np.ndarray]) -> Tuple[torch.Tensor, np.ndarray]:
        """Padded uint8 batch [B, Hm, Wm, 3] on the device (Hm, Wm: multiples of 32, so that a stream of similar
        images re-uses few shapes) and the images' own (h, w)."""
        sizes = np.array([im.shape[:2] for im in images], dtype=np.int32)
        hm = int(-(-int(sizes[:, 0].max()) // 32) * 32)
        wm = int(-(-int(sizes[:, 1].max()) // 32) * 32)
        host = torch.empty((len(images), hm, wm, 3), dtype=torch.uint8, pin_memory=self.on_gpu)
        view = host.numpy()
        for i, im in enumerate(images):
            h, w = im.shape[:2]
            view[i, :h, :w] = im
            view[i, h:] = 0          # only the padding is cleared (nothing of it reaches a result: the gathers stop at
            view[i, :h, w:] = 0      # the image's own extent; a whole-canvas fill was the slowest step of a batch)
        return host.to(self.device, non_blocking=True), sizes

    def _h2d(self, arr: np.ndarray) -> torch.Tensor:
        """Small host array -> device through pinned memory: a copy from pageable memory blocks the host until the
        stream has drained (profiles/r3_instance_loader_sync.md), once per batch."""
        t = torch.from_numpy(np.ascontiguousarray(arr))
        return t.pin_memory().to(self.device, non_blocking=True) if self.on_gpu else t

    @torch.no_grad()
    def _forward(self, x: torch.Tensor) -> torch.Tensor:
        out = self._fwd(x)
        return out[0] if isinstance(out, (tuple, list)) else out


class SemSegPredictor(_Base):
    """``predict(images)``: uint8 class maps, one per image, at the image's own size.

    ``fit``: 'letterbox' keeps the aspect ratio (the longer side becomes ``crop``, the rest of the square is the mean
    colour, i.e. zeros for the network — what the training crops pad with) or 'stretch' (both sides become ``crop``)."""

    def __init__(self, model: nn.Module, device=None, crop: int = 513, fit: str = "letterbox", batch: int = 8,
                 graph: bool = True):
        super().__init__(model, device, graph)
        if fit not in ("letterbox", "stretch"):
            raise ValueError("fit: 'letterbox' or 'stretch'")
        self.crop, self.fit, self.batch = int(crop), fit, max(int(batch), 1)

    def _extent(self, h: int, w: int) -> Tuple[int, int]:
        if self.fit == "stretch":
            return self.crop, self.crop
        s = self.crop / float(max(h, w))
        return max(1, min(self.crop, int(round(h * s)))), max(1, min(self.crop, int(round(w * s))))

    @torch.no_grad()
    def logits(self, images: Sequence[np.ndarray]) -> List[torch.Tensor]:
        """fp32 logits [C, h, w] per image, on the device (all of them at once: 16 MB per 375x500 image of 21 classes —
        ``predict`` reduces each batch to label maps before the next one runs)."""
        return list(self._logits_iter(images))

    def _logits_iter(self, images: Sequence[np.ndarray]):
        images = [_as_u8_rgb(im) for im in images]
        for s0 in range(0, len(images), self.batch):
            chunk = images[s0:s0 + self.batch]
            n = len(chunk)
            if self.on_gpu and n < self.batch and s0 > 0:
                chunk = chunk + [chunk[-1]] * (self.batch - n)      # the tail re-uses the captured batch shape
            canvas, sizes = self._canvas(chunk)
            ext = [self._extent(int(h), int(w)) for h, w in sizes]
            mats = np.zeros((len(chunk), 6), dtype=np.float32)
            for i, ((h, w), (eh, ew)) in enumerate(zip(sizes, ext)):
                mats[i, 0] = (w - 1) / max(ew - 1, 1)               # align-corners map of the image onto its extent
                mats[i, 4] = (h - 1) / max(eh - 1, 1)
            x, _ = affine_augment(canvas, None, self._h2d(mats),
                                  (self.crop, self.crop), out_dtype=self.dtype)
            if self.fit == "letterbox":
                # exactly zero (the mean colour, what training crops are padded with) beyond each image's extent: a result
                # must not depend on what shares the batch and sizes the canvas
                e = self._h2d(np.asarray(ext, dtype=np.int32))
                ar = torch.arange(self.crop, device=self.device, dtype=torch.int32)
                keep = (ar.view(1, -1, 1) < e[:, 0].view(-1, 1, 1)) & (ar.view(1, 1, -1) < e[:, 1].view(-1, 1, 1))
                x.mul_(keep.unsqueeze(1))
            lg = self._forward(x)
            for i in range(n):
                h, w = int(sizes[i][0]), int(sizes[i][1])
                eh, ew = ext[i]
                yield F.interpolate(lg[i:i + 1, :, :eh, :ew], size=(h, w), mode="bilinear",
                                    align_corners=True)[0].float()

    @torch.no_grad()
    def predict(self, images: Sequence[np.ndarray]) -> List[np.ndarray]:
        res = []
        for lg in self._logits_iter(images):
            cls = (lg[0] > 0).to(torch.uint8) if lg.shape[0] == 1 else lg.argmax(0).to(torch.uint8)
            if self.on_gpu:          # into pinned memory without waiting: ONE synchronisation for the whole call
                host = torch.empty(cls.shape, dtype=torch.uint8, pin_memory=True)
                host.copy_(cls, non_blocking=True)
                cls = host
            res.append(cls)
        if self.on_gpu:
            torch.cuda.current_stream(self.device).synchronize()
        return [c.numpy() for c in res]


class InstancePredictor(_Base):
    """``predict(images, points)``: float32 probability maps of the objects whose extreme points were given.

    ``points``: per image four (x, y) pixel positions (left-most, right-most, top and bottom point of the object, in
    any order).  The crop is the box of the points grown by ``relax`` pixels (zero-padded outside the image when
    ``zero_pad``), as ``CropFromMaskStatic`` makes it from the ground-truth mask during validation."""

    def __init__(self, model: nn.Module, device=None, crop: int = 512, relax: int = 50, zero_pad: bool = True,
                 alpha: float = 0.6, batch: int = 8, graph: bool = True):
        super().__init__(model, device, graph)
        self.crop, self.relax, self.zero_pad, self.alpha = int(crop), int(relax), bool(zero_pad), float(alpha)
        self.batch = max(int(batch), 1)

    @torch.no_grad()
    def network_input(self, images: Sequence[np.ndarray], points) -> Tuple[torch.Tensor, np.ndarray, np.ndarray]:
        """(concat [B,4,crop,crop] channels-last with values 0..255, crop boxes int [B,4], sizes [B,2])."""
        images = [_as_u8_rgb(im) for im in images]
        pts = np.asarray(points, dtype=np.float32).reshape(len(images), 4, 2)
        canvas, sizes = self._canvas(images)
        for (h, w), p in zip(sizes, pts):
            if p[:, 0].min() < 0 or p[:, 1].min() < 0 or p[:, 0].max() > w - 1 or p[:, 1].max() > h - 1:
                raise ValueError(f"extreme points outside their {h}x{w} image: {p.tolist()}")
        bbox = np.stack([np.floor(pts[:, :, 0].min(1)), np.floor(pts[:, :, 1].min(1)),
                         np.ceil(pts[:, :, 0].max(1)), np.ceil(pts[:, :, 1].max(1))], axis=1).astype(np.int32)
        dev = self.device
        n = len(images)
        sizes_t = self._h2d(sizes)
        ident = _identity_mats(n, dev) if self.on_gpu else torch.tensor([[1., 0, 0, 0, 1, 0]]).repeat(n, 1)
        mats, clip, _ = compose_crop(ident, self._h2d(bbox), (self.crop, self.crop), self.relax, self.zero_pad, sizes_t,
                                     tuple(canvas.shape[1:3]))
        idx = torch.arange(n, dtype=torch.int32, device=dev)
        x, _ = affine_augment2(canvas, None, mats, (self.crop, self.crop), idx, sizes_t, clip, interp="bicubic",
                               clamp=True, mean=(0.0, 0.0, 0.0), std=(1.0, 1.0, 1.0), border=(0.0, 0.0, 0.0),
                               channels=4, out_dtype=self.dtype, validated=True)
        # the relaxed box, as compose_crop makes it — computed here as well so that the host need not read it back
        cb = bbox.astype(np.int64) + np.array([-self.relax, -self.relax, self.relax, self.relax], dtype=np.int64)
        if not self.zero_pad:
            cb[:, 0], cb[:, 1] = np.maximum(cb[:, 0], 0), np.maximum(cb[:, 1], 0)
            cb[:, 2], cb[:, 3] = np.minimum(cb[:, 2], sizes[:, 1] - 1), np.minimum(cb[:, 3], sizes[:, 0] - 1)
        # the clicked points in crop coordinates (pixel centres of the resize: out = (src + 0.5) / scale - 0.5)
        sx = (cb[:, 2] - cb[:, 0] + 1) / float(self.crop)
        sy = (cb[:, 3] - cb[:, 1] + 1) / float(self.crop)
        pc = np.empty_like(pts)
        pc[:, :, 0] = (pts[:, :, 0] - cb[:, None, 0] + 0.5) / sx[:, None] - 0.5
        pc[:, :, 1] = (pts[:, :, 1] - cb[:, None, 1] + 0.5) / sy[:, None] - 0.5
        pc = np.clip(np.rint(pc), 0, self.crop - 1).astype(np.int32)
        nellipse_guidance_(x, self._h2d(pc), 3, self.alpha)
        return x, cb, sizes

    def _paste_back_device(self, prob: torch.Tensor, cb: np.ndarray, sizes: np.ndarray) -> List[np.ndarray]:
        """``helpers.crop2fullmask`` on the device: the crop-sized probabilities resized (bicubic, pixel centres, the
        a = -0.75 kernel of the host form) to their relaxed boxes, the relax border suppressed, the part inside the image
        written into a canvas of the image's size; results leave through pinned memory, one synchronisation."""
        out = []
        for i in range(prob.shape[0]):
            h, w = int(sizes[i][0]), int(sizes[i][1])
            x0, y0, x1, y1 = (int(v) for v in cb[i])
            bh, bw = y1 - y0 + 1, x1 - x0 + 1
            full = torch.zeros((h, w), dtype=torch.float32, device=self.device)
            vx0, vy0, vx1, vy1 = max(x0, 0), max(y0, 0), min(x1, w - 1), min(y1, h - 1)
            if bh > 0 and bw > 0 and vx1 >= vx0 and vy1 >= vy0:
                c = F.interpolate(prob[i][None, None], size=(bh, bw), mode="bicubic", align_corners=False)[0, 0]
                r = self.relax
                if r > 0:
                    keep = torch.zeros_like(c)
                    keep[r:bh - r, r:bw - r] = 1.0
                    c = c * keep
                full[vy0:vy1 + 1, vx0:vx1 + 1] = c[vy0 - y0:vy1 - y0 + 1, vx0 - x0:vx1 - x0 + 1]
            host = torch.empty((h, w), dtype=torch.float32, pin_memory=True)
            host.copy_(full.clamp_(0.0, 1.0), non_blocking=True)
            out.append(host)
        torch.cuda.current_stream(self.device).synchronize()
        return [t.numpy() for t in out]

    @torch.no_grad()
    def predict(self, images: Sequence[np.ndarray], points) -> List[np.ndarray]:
        from .data import helpers
        pts = np.asarray(points, dtype=np.float32).reshape(len(images), 4, 2)
        res: List[np.ndarray] = []
        for s0 in range(0, len(images), self.batch):
            chunk, cp = list(images[s0:s0 + self.batch]), pts[s0:s0 + self.batch]
            n = len(chunk)
            if self.on_gpu and n < self.batch and s0 > 0:
                chunk, cp = chunk + [chunk[-1]] * (self.batch - n), np.concatenate([cp, np.repeat(cp[-1:], self.batch - n, 0)])
            x, cb, sizes = self.network_input(chunk, cp)
            if self.on_gpu:
                res.extend(self._paste_back_device(torch.sigmoid(self._forward(x)[:n, 0].float()), cb, sizes))
                continue
            prob = torch.sigmoid(self._forward(x)[:n, 0].float()).cpu().numpy()       # one host read per batch
            for i in range(n):
                full = helpers.crop2fullmask(prob[i], tuple(int(v) for v in cb[i]),
                                             im_size=(int(sizes[i][0]), int(sizes[i][1])),
                                             zero_pad=self.zero_pad, relax=self.relax)
                res.append(np.clip(full, 0.0, 1.0, out=full))          # (the bicubic paste-back overshoots)
        return res


def load_predictor(task: str, model_name: str, weights: Optional[str] = None, num_classes: Optional[int] = None,
                   output_stride: Optional[int] = None, device=None, **kw):
    """Model by name (+ a checkpoint written by this framework or by the reference: utils.checkpoint.load_weights) in
    the predictor of ``task`` ('semseg' | 'instance')."""
    from .models import build_model
    from .utils.checkpoint import load_weights
    instance = task == "instance"
    nclass = num_classes if num_classes is not None else (1 if instance else 21)
    model = build_model(model_name, nclass, output_stride=output_stride, in_channels=4 if instance else 3)
    if weights:
        load_weights(weights, model, strict=True)
    cls = InstancePredictor if instance else SemSegPredictor
    return cls(model, device, **kw)


__all__ = ["SemSegPredictor", "InstancePredictor", "load_predictor", "voc_palette"]
