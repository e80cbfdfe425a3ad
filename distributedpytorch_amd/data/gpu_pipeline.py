"""Device-side data path: the loader hands the model augmented batches that were produced ON the GPU.

The reference augments on two CPU DataLoader workers and copies float batches synchronously
(`/root/reference/train_pascal.py:123-134`, `:161`, `:193`; its TODO `:5` asks for non-blocking
copies).  A 950 img/s consumer cannot be fed that way.  MI355X has 288 GB of HBM: the whole DECODED
dataset (VOC+SBD: 10,582 images ≈ 8 GB as uint8) fits beside the model many times over, so

* ``DeviceSampleStore`` holds every sample once, uint8, padded to a common slot size, on the GPU
  (filled once at start-up by DataLoader workers that only decode — no CPU resize / warp /
  normalise — through pinned buffers and a side stream);
* per step the loader draws the affine parameters on the host (a few hundred floats), and ONE
  gather kernel (``ops.affine_augment2``: scale + rotate + crop + flip + normalise, NHWC bf16 out)
  produces the batch from the store by slot index — steady state has no host data path at all;
* the instance task runs the reference's whole chain (flip → scale/rotate → mask crop → resize →
  n-ellipse guidance → 4-channel concat) the same way (``ops.instance_augment``);
* datasets larger than the HBM budget stream instead: raw uint8 batches, pinned, copied one batch
  ahead on a side stream, augmented on arrival (``mode='stream'``).

The CPU transform path (``data.transforms``) stays selectable (``train_pascal.py --data-path cpu``)
for behaviour parity with the reference.
"""
from __future__ import annotations

from typing import Dict, Iterator, Optional, Tuple

import numpy as np
import torch
from torch.utils.data import DataLoader, Dataset

from .. import ops

MEAN_U8 = (124, 116, 104)      # uint8 rounding of the ImageNet mean colour: the image padding of a slot


class RawUint8(Dataset):
    """Wraps a dataset built with ``transform=None`` and yields what the device store keeps: uint8 HWC image,
    uint8 label (semantic: class map with 255 = ignore; instance: binary object mask) and, for the instance task,
    the void mask — nothing is resized or normalised on the host."""

    def __init__(self, base: Dataset, task: str = "semseg"):
        self.base, self.task = base, task

    def __len__(self) -> int:
        return len(self.base)

    def __getitem__(self, i: int) -> Dict[str, torch.Tensor]:
        s = self.base[i]
        img = np.asarray(s["image"])
        gt = np.asarray(s["gt"])
        out = {"image": torch.from_numpy(np.ascontiguousarray(np.clip(img, 0, 255).astype(np.uint8))),
               "label": torch.from_numpy(np.ascontiguousarray(gt.astype(np.uint8))),
               "index": torch.tensor(i, dtype=torch.int64)}
        if "void_pixels" in s:
            out["void"] = torch.from_numpy(np.ascontiguousarray(np.asarray(s["void_pixels"]).astype(np.uint8)))
        return out


def pad_collate(samples, slot: Optional[Tuple[int, int]] = None, label_fill: int = 255):
    """Stack samples of different sizes into [B, Hs, Ws(,3)] slots: image padding = mean colour, label padding =
    ``label_fill``; the true sizes travel along."""
    hs = max(int(s["image"].shape[0]) for s in samples)
    ws = max(int(s["image"].shape[1]) for s in samples)
    if slot is not None:
        if hs > slot[0] or ws > slot[1]:
            raise ValueError(f"sample {hs}x{ws} larger than the slot {slot}")
        hs, ws = slot
    n = len(samples)
    img = torch.empty(n, hs, ws, 3, dtype=torch.uint8)
    img[...] = torch.tensor(MEAN_U8, dtype=torch.uint8)
    lab = torch.full((n, hs, ws), label_fill, dtype=torch.uint8)
    void = torch.zeros(n, hs, ws, dtype=torch.uint8) if "void" in samples[0] else None
    sizes = torch.empty(n, 2, dtype=torch.int32)
    idx = torch.empty(n, dtype=torch.int64)
    for k, s in enumerate(samples):
        h, w = int(s["image"].shape[0]), int(s["image"].shape[1])
        img[k, :h, :w] = s["image"]
        lab[k, :h, :w] = s["label"]
        if void is not None:
            void[k, :h, :w] = s["void"]
        sizes[k, 0], sizes[k, 1] = h, w
        idx[k] = s["index"]
    out = {"image": img, "label": lab, "sizes": sizes, "index": idx}
    if void is not None:
        out["void"] = void
    return out


class DeviceSampleStore:
    """The decoded dataset, resident on the device: ``images`` uint8 [N,Hs,Ws,3], ``labels`` uint8 [N,Hs,Ws],
    ``sizes`` int32 [N,2]."""

    def __init__(self, n: int, slot: Tuple[int, int], device: torch.device, label_fill: int = 255,
                 keep_void: bool = False):
        self.n, self.slot, self.device = n, tuple(slot), device
        self.images = torch.empty(n, slot[0], slot[1], 3, dtype=torch.uint8, device=device)
        self.labels = torch.full((n, slot[0], slot[1]), label_fill, dtype=torch.uint8, device=device)
        # void masks (instance validation: pixels excluded from the Jaccard index, train_pascal.py:291)
        self.voids = torch.zeros(n, slot[0], slot[1], dtype=torch.uint8, device=device) if keep_void else None
        self.sizes = torch.ones(n, 2, dtype=torch.int32, device=device)
        self.sizes_host = np.ones((n, 2), dtype=np.int32)
        self.filled = np.zeros(n, dtype=bool)

    @property
    def nbytes(self) -> int:
        return self.images.numel() + self.labels.numel() + (self.voids.numel() if self.voids is not None else 0)

    @staticmethod
    def bytes_needed(n: int, slot: Tuple[int, int]) -> int:
        return n * slot[0] * slot[1] * 4

    def fill(self, raw: Dataset, workers: int = 4, batch: int = 32, label_fill: int = 255, progress=None) -> None:
        """Decode every sample once (workers) and copy it to its slot through pinned memory on a side stream."""
        on_gpu = self.device.type == "cuda"
        loader = DataLoader(raw, batch_size=batch, shuffle=False, num_workers=workers, pin_memory=on_gpu,
                            collate_fn=lambda ss: pad_collate(ss, self.slot, label_fill))
        side = torch.cuda.Stream(device=self.device) if on_gpu else None
        for k, b in enumerate(loader):
            idx = b["index"]
            if on_gpu:
                with torch.cuda.stream(side):
                    di = idx.to(self.device, non_blocking=True)
                    self.images.index_copy_(0, di, b["image"].to(self.device, non_blocking=True))
                    self.labels.index_copy_(0, di, b["label"].to(self.device, non_blocking=True))
                    if self.voids is not None and "void" in b:
                        self.voids.index_copy_(0, di, b["void"].to(self.device, non_blocking=True))
                    self.sizes.index_copy_(0, di, b["sizes"].to(self.device, non_blocking=True))
                side.synchronize()          # the pinned batch is recycled by the loader after this iteration
            else:
                self.images[idx], self.labels[idx], self.sizes[idx] = b["image"], b["label"], b["sizes"]
                if self.voids is not None and "void" in b:
                    self.voids[idx] = b["void"]
            self.sizes_host[idx.numpy()] = b["sizes"].numpy()
            self.filled[idx.numpy()] = True
            if progress is not None:
                progress(k, len(loader))
        if not self.filled.all():
            raise RuntimeError("device sample store: some samples were never delivered")


def _max_size(raw: Dataset, probe: int = 0) -> Tuple[int, int]:
    base = getattr(raw, "base", raw)
    size = getattr(base, "size", None)              # SyntheticVOC
    if size is not None:
        return int(size[0]), int(size[1])
    return 512, 512                                  # VOC2012 images are at most 500 x 500


class GpuBatchLoader:
    """Iterable of ``(x, y)`` device batches for one rank.

    ``task='semseg'``: x = normalised RGB [B,3,crop,crop] channels-last bf16 (fp32 on CPU), y = uint8 class map
    with 255 = ignore; multi-scale (``scales``), random crop, flip — `BASELINE.json` config 5's "multi-scale
    augmentation".  ``task='instance'``: x = [B,4,crop,crop] RGB + guidance in 0..255, y = float {0,1}
    [B,1,crop,crop] (the tensors the reference's in-loop asserts describe, `train_pascal.py:188-190`)."""

    def __init__(self, base: Dataset, sampler, batch: int, device: torch.device, crop: int, task: str = "semseg",
                 scales=(0.5, 2.0), rots=(0.0, 0.0), flip_p: float = 0.5, seed: int = 0, rank: int = 0,
                 workers: int = 4, mode: str = "auto", hbm_budget_gb: float = 64.0, train: bool = True,
                 out_dtype: Optional[torch.dtype] = None, relax: int = 50, zero_pad: bool = True,
                 drop_last: Optional[bool] = None):
        self.raw = RawUint8(base, task)
        self.sampler, self.batch, self.device, self.crop, self.task = sampler, int(batch), device, int(crop), task
        self.scales, self.rots, self.flip_p, self.train = scales, rots, flip_p, train
        self.relax, self.zero_pad = relax, zero_pad
        self.drop_last = train if drop_last is None else bool(drop_last)     # validation visits the tail batch too
        self.with_meta = False               # iter_eval(): batches come with what the device metric needs
        self.workers = workers
        self.out_dtype = out_dtype or (torch.bfloat16 if device.type == "cuda" else torch.float32)
        self.rng = np.random.default_rng([seed, rank, 0x9E37])       # rank-dependent augmentation stream
        self.label_fill = 255 if task == "semseg" else 0
        self.slot = _max_size(self.raw)
        need = DeviceSampleStore.bytes_needed(len(self.raw), self.slot)
        if mode == "auto":
            mode = "cache" if need <= hbm_budget_gb * 2 ** 30 else "stream"
        self.mode = mode
        self.store: Optional[DeviceSampleStore] = None
        self.epoch = 0
        self.steps = 0

    def __len__(self) -> int:
        n = len(self.sampler)
        return n // self.batch if self.drop_last else (n + self.batch - 1) // self.batch

    def set_epoch(self, epoch: int) -> None:
        self.epoch = epoch
        if hasattr(self.sampler, "set_epoch"):
            self.sampler.set_epoch(epoch)

    def state_dict(self) -> dict:
        """This rank's augmentation stream (ints as strings: PCG64's words are 128 bits wide), the sampler epoch and
        the step count — what ``utils.checkpoint.save_checkpoint(rank_state=...)`` stores per rank."""
        bg = self.rng.bit_generator.state
        return {"kind": "gpu", "bit_generator": str(bg["bit_generator"]), "state": str(bg["state"]["state"]),
                "inc": str(bg["state"]["inc"]), "has_uint32": int(bg["has_uint32"]), "uinteger": int(bg["uinteger"]),
                "epoch": int(self.epoch), "steps": int(self.steps)}

    def load_state_dict(self, st: dict) -> None:
        if not st or st.get("kind") != "gpu":
            return
        if st["bit_generator"] != self.rng.bit_generator.state["bit_generator"]:
            raise ValueError(f"loader state was written by a {st['bit_generator']} generator")
        self.rng.bit_generator.state = {"bit_generator": st["bit_generator"],
                                        "state": {"state": int(st["state"]), "inc": int(st["inc"])},
                                        "has_uint32": int(st["has_uint32"]), "uinteger": int(st["uinteger"])}
        self.steps = int(st.get("steps", 0))
        self.set_epoch(int(st.get("epoch", 0)))

    def reseed(self, seed: int, rank: int, salt: int) -> None:
        """A fresh rank-dependent stream (a restore after a non-finite loss must not replay the same batches)."""
        self.rng = np.random.default_rng([seed, rank, 0x9E37, salt])

    def prepare(self, progress=None) -> None:
        if self.mode == "cache" and self.store is None:
            self.store = DeviceSampleStore(len(self.raw), self.slot, self.device, self.label_fill,
                                           keep_void=self.task == "instance" and not self.train)
            self.store.fill(self.raw, self.workers, label_fill=self.label_fill, progress=progress)

    # ------------------------------------------------------------------
    def iter_eval(self):
        """Validation batches ``(x, y, meta)``: ``meta`` = what ``ops.augment.paste_jaccard`` needs to score the
        predictions on the device — the full-size masks (``gt``, ``void``: the store's tensors or the uploaded
        batch), ``src_idx`` into them, ``sizes`` and, for the instance task, the ``crop_bbox`` of every object."""
        self.with_meta = True
        try:
            yield from iter(self)
        finally:
            self.with_meta = False

    def _augment(self, images, labels, src_idx, sizes_dev, sizes_host, idx_host=None, voids=None):
        bsz = len(sizes_host)
        # validate the HOST copies; the device path then launches without reading anything back (no host sync)
        ops.augment.check_slots(idx_host, sizes_host, images.shape[0], images.shape[1:3])
        out = (self.crop, self.crop)
        self.steps += 1
        if self.task == "semseg":
            if self.train:
                mats = ops.make_affine_params(bsz, sizes_host, out, self.scales, self.rots, self.flip_p, self.rng)
            else:           # validation: short side to the crop size, centre crop (transforms.FixedScaleCenterCrop)
                mats = np.zeros((bsz, 6), dtype=np.float32)
                for i, (h, w) in enumerate(sizes_host):
                    s = max(self.crop / h, self.crop / w)
                    mats[i] = (1 / s, 0, ((w * s - self.crop) / 2 + 0.5) / s - 0.5,
                               0, 1 / s, ((h * s - self.crop) / 2 + 0.5) / s - 0.5)
            # a few hundred bytes from pageable memory: the runtime stages such copies itself; pinning a fresh buffer
            # per step costs a host allocation of milliseconds (measured as 12 ms of data wait per step)
            m = torch.from_numpy(mats).to(self.device, non_blocking=True)
            x, y = ops.affine_augment2(images, labels, m, out, src_idx, sizes_dev, out_dtype=self.out_dtype,
                                       validated=True)
            return (x, y, {"gt": labels, "void": None, "src_idx": src_idx, "sizes": sizes_dev}) if self.with_meta \
                else (x, y)
        warp = ops.make_warp_params(bsz, sizes_host, self.rots, self.scales, self.flip_p if self.train else 0.0,
                                    self.rng) if self.train else \
            np.tile(np.array([1, 0, 0, 0, 1, 0], dtype=np.float32), (bsz, 1))
        w = torch.from_numpy(np.ascontiguousarray(warp, dtype=np.float32)).to(self.device, non_blocking=True)
        seed = int(self.rng.integers(0, 2 ** 31 - 1)) if self.train else 0     # validation draws nothing
        x, y, bbox = ops.instance_augment(images, labels, src_idx, sizes_dev, w, out, self.relax, self.zero_pad,
                                          random_points=self.train, seed=seed, out_dtype=self.out_dtype, validated=True)
        if self.with_meta:
            return x, y, {"gt": labels, "void": voids, "src_idx": src_idx, "sizes": sizes_dev, "crop_bbox": bbox}
        return x, y

    def __iter__(self) -> Iterator[Tuple[torch.Tensor, torch.Tensor]]:
        if self.mode == "cache":
            yield from self._iter_cache()
        else:
            yield from self._iter_stream()

    def _iter_cache(self):
        self.prepare()
        st = self.store
        order = np.fromiter(iter(self.sampler), dtype=np.int64)
        for k in range(len(self)):
            idx = order[k * self.batch:(k + 1) * self.batch]
            src = torch.from_numpy(idx.astype(np.int32)).to(self.device, non_blocking=True)
            sizes_host = st.sizes_host[idx]
            sizes_dev = st.sizes.index_select(0, src.long()) if self.device.type == "cuda" else st.sizes[idx]
            yield self._augment(st.images, st.labels, src, sizes_dev, sizes_host, idx, st.voids)

    def _iter_stream(self):
        on_gpu = self.device.type == "cuda"
        loader = DataLoader(self.raw, batch_size=self.batch, sampler=self.sampler, num_workers=self.workers,
                            drop_last=self.drop_last, pin_memory=on_gpu, persistent_workers=False,
                            collate_fn=lambda ss: pad_collate(ss, None, self.label_fill))
        # ONE copy stream per loader, not one per epoch: the caching allocator keeps a pool per stream, PyTorch hands
        # out its 32 pool streams round-robin, so a fresh stream every epoch allocated its upload buffers anew each
        # time and reserved memory grew by half a GiB per epoch (B128: 45.0 -> 50.5 GiB in 10 epochs,
        # profiles/r3_stream_mode_reserved_growth.md)
        if on_gpu and getattr(self, "_side_stream", None) is None:
            self._side_stream = torch.cuda.Stream(device=self.device)
        side = self._side_stream if on_gpu else None

        def upload(b):
            if not on_gpu:
                return b, None
            with torch.cuda.stream(side):
                d = {k: v.to(self.device, non_blocking=True) for k, v in b.items() if k != "index"}
                ev = torch.cuda.Event()
                ev.record(side)
            d["_sizes_host"] = b["sizes"].numpy().copy()
            return d, ev

        it = iter(loader)
        try:
            nxt = upload(next(it))
        except StopIteration:
            return
        while nxt is not None:
            cur, ev = nxt
            try:
                nxt = upload(next(it))          # one batch ahead: its copy overlaps this batch's step
            except StopIteration:
                nxt = None
            if ev is not None:
                torch.cuda.current_stream(self.device).wait_event(ev)
                for v in cur.values():
                    if torch.is_tensor(v):
                        v.record_stream(torch.cuda.current_stream(self.device))
            sizes_host = cur.pop("_sizes_host") if on_gpu else cur["sizes"].numpy()
            yield self._augment(cur["image"], cur["label"], None, cur["sizes"], sizes_host, None, cur.get("void"))
