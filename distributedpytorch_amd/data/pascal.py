"""PASCAL VOC 2012 segmentation dataset — public surface of `/root/reference/pascal.py`.

``VOCSegmentation`` keeps the reference's constructor (``root, split, transform, download,
preprocess, area_thres, retname, suppress_void_pixels, default``, `pascal.py:32-41`), its
*one item = one object instance* indexing (`:107-116`, `:139-140`), the sample-dict contract

    image        float32 H×W×3 RGB 0..255            (`:237`)
    gt           float32 H×W  {0,1}  (instance ids when ``default=True``)   (`:247-251`)
    void_pixels  float32 H×W  {0,1}                   (`:241`, `:261`)
    meta         {'image', 'object', 'category', 'im_size'}   (`:126-132`)

the ``<splits>_instances[_area_thres-N].txt`` JSON cache (`:60-65`, `:154-195`) and ``__str__``
(`:265-266`).  Differences, all deliberate:

* ``semantic=True`` selects the north-star mode the reference lacks: one item = one IMAGE,
  ``gt`` = 21-class map with 255 = ignore (from ``SegmentationClass``), keys ``image``/``gt``/``meta``.
* integrity: the tar MD5 is streamed in 1 MiB blocks (the reference reads the whole 2 GB file
  into memory on every construction, `:147`) and ``check_integrity=False`` accepts an already
  extracted tree.
* multi-rank safety: the cache file is written by rank 0 only, atomically, under
  ``dist_env.master_first`` (the reference's plain ``open(...,'w')`` races under several ranks).
* ``np.int`` (`:254`, removed in NumPy ≥ 1.24) is gone.

``SyntheticVOC`` is a VOC-shaped random dataset (513×513-style RGB, 21 classes + 255 void) for
benchmarks and tests — no network access is available to fetch the real archive.
"""
from __future__ import annotations

import hashlib
import json
import os
import tarfile
from typing import Callable, Dict, List, Optional, Sequence, Union

import numpy as np
import torch.utils.data as data
from PIL import Image

from ..parallel import dist_env

try:  # `from mypath import Path` in the reference (`pascal.py:13`)
    from mypath import Path as _Path
    _DEFAULT_ROOT = _Path.db_root_dir("pascal")
except Exception:  # pragma: no cover - mypath is optional
    _DEFAULT_ROOT = os.environ.get("VOC_ROOT", "./data/pascal")

VOC_CLASSES = ("background", "aeroplane", "bicycle", "bird", "boat", "bottle", "bus", "car", "cat", "chair",
               "cow", "diningtable", "dog", "horse", "motorbike", "person", "pottedplant", "sheep", "sofa",
               "train", "tvmonitor")
VOID_LABEL = 255


class VOCSegmentation(data.Dataset):
    URL = "http://host.robots.ox.ac.uk/pascal/VOC/voc2012/VOCtrainval_11-May-2012.tar"
    FILE = "VOCtrainval_11-May-2012.tar"
    MD5 = "6cd6e144f989b92b3379bac3b3de84fd"
    BASE_DIR = "VOCdevkit/VOC2012"
    category_names = list(VOC_CLASSES)

    def __init__(self, root: str = _DEFAULT_ROOT, split: Union[str, Sequence[str]] = "val",
                 transform: Optional[Callable] = None, download: bool = False, preprocess: bool = False,
                 area_thres: int = 0, retname: bool = True, suppress_void_pixels: bool = True,
                 default: bool = False, semantic: bool = False, check_integrity: bool = True):
        self.root = root
        self.transform = transform
        self.split = [split] if isinstance(split, str) else sorted(split)
        self.area_thres = area_thres
        self.retname = retname
        self.suppress_void_pixels = suppress_void_pixels
        self.default = default
        self.semantic = semantic
        voc = os.path.join(root, self.BASE_DIR)
        self._dirs = {"img": os.path.join(voc, "JPEGImages"), "cls": os.path.join(voc, "SegmentationClass"),
                      "obj": os.path.join(voc, "SegmentationObject"),
                      "sets": os.path.join(voc, "ImageSets", "Segmentation")}
        suffix = f"_area_thres-{area_thres}" if area_thres != 0 else ""
        self.obj_list_file = os.path.join(self._dirs["sets"], "_".join(self.split) + "_instances" + suffix + ".txt")

        if download:
            self._download()
        if check_integrity and not self._check_integrity():
            raise RuntimeError("Dataset not found or corrupted. You can use download=True to download it")

        self.im_ids: List[str] = []
        self.images: List[str] = []
        self.categories: List[str] = []
        self.masks: List[str] = []
        for name in self.split:
            with open(os.path.join(self._dirs["sets"], name + ".txt")) as fh:
                ids = [ln.strip() for ln in fh.read().splitlines() if ln.strip()]
            for im_id in ids:
                paths = (os.path.join(self._dirs["img"], im_id + ".jpg"),
                         os.path.join(self._dirs["cls"], im_id + ".png"),
                         os.path.join(self._dirs["obj"], im_id + ".png"))
                for p in paths:
                    if not os.path.isfile(p):
                        raise FileNotFoundError(p)
                self.im_ids.append(im_id)
                self.images.append(paths[0])
                self.categories.append(paths[1])
                self.masks.append(paths[2])

        # per-image object categories (cached JSON); rank 0 builds it, the others read it
        with dist_env.master_first():
            if preprocess and dist_env.is_master():
                self._preprocess()
            elif not self._check_preprocess():
                if dist_env.is_master():
                    print("Preprocessing of PASCAL VOC dataset, this will take long, but it will be done only once.")
                    self._preprocess()
                else:  # pragma: no cover - reached only if rank 0 failed
                    raise RuntimeError("object cache missing after rank 0 preprocessing")
        self.obj_list = [[ii, jj] for ii, im_id in enumerate(self.im_ids)
                         for jj, cat in enumerate(self.obj_dict[im_id]) if cat != -1]
        n_img = len({ii for ii, _ in self.obj_list})
        if dist_env.is_master():
            print("Number of images: {:d}\nNumber of objects: {:d}".format(n_img, len(self.obj_list)))

    # ------------------------------------------------------------------ Dataset protocol
    def __len__(self) -> int:
        return len(self.im_ids) if self.semantic else len(self.obj_list)

    def __getitem__(self, index: int) -> Dict:
        if self.semantic:
            sample = self._semantic_item(index)
        else:
            sample = self._instance_item(index)
        return self.transform(sample) if self.transform is not None else sample

    def __str__(self) -> str:
        return "VOC2012(split=" + str(self.split) + ",area_thres=" + str(self.area_thres) + ")"

    # ------------------------------------------------------------------ items
    def _load_rgb(self, ii: int) -> np.ndarray:
        return np.asarray(Image.open(self.images[ii]).convert("RGB"), dtype=np.float32)

    def _semantic_item(self, ii: int) -> Dict:
        img = self._load_rgb(ii)
        cls = np.asarray(Image.open(self.categories[ii]), dtype=np.float32)
        sample = {"image": img, "gt": cls}
        if self.retname:
            sample["meta"] = {"image": str(self.im_ids[ii]), "im_size": (img.shape[0], img.shape[1])}
        return sample

    def _instance_item(self, index: int) -> Dict:
        img, target, void = self._make_img_gt_point_pair(index)[:3]
        sample = {"image": img, "gt": target, "void_pixels": void}
        if self.retname:
            ii, jj = self.obj_list[index]
            sample["meta"] = {"image": str(self.im_ids[ii]), "object": str(jj),
                              "category": self.obj_dict[self.im_ids[ii]][jj],
                              "im_size": (img.shape[0], img.shape[1])}
        return sample

    def _make_img_gt_point_pair(self, index: int):
        """(image, target, void, other_classes, other_same_class, background) as float32 maps."""
        ii, jj = self.obj_list[index]
        img = self._load_rgb(ii)
        inst = np.asarray(Image.open(self.masks[ii]), dtype=np.float32).copy()
        void = inst == VOID_LABEL
        inst[void] = 0
        background = (inst == 0) & ~void
        other_same = np.zeros(inst.shape, dtype=bool)
        other_diff = np.zeros(inst.shape, dtype=bool)
        if self.default:
            target = inst
        else:
            target = (inst == jj + 1).astype(np.float32)
            cats = self.obj_dict[self.im_ids[ii]]
            for k in range(1, int(inst.max()) + 1):
                if k == jj + 1 or k - 1 >= len(cats):
                    continue
                if cats[k - 1] == cats[jj]:
                    other_same |= inst == k
                else:
                    other_diff |= inst == k
        f32 = np.float32
        return img, target, void.astype(f32), other_diff.astype(f32), other_same.astype(f32), background.astype(f32)

    # ------------------------------------------------------------------ cache of object categories
    def _check_preprocess(self) -> bool:
        if not os.path.isfile(self.obj_list_file):
            return False
        with open(self.obj_list_file) as fh:
            self.obj_dict = json.load(fh)
        return sorted(str(k) for k in self.obj_dict.keys()) == sorted(self.im_ids)

    def _preprocess(self) -> None:
        self.obj_dict = {}
        for ii, im_id in enumerate(self.im_ids):
            inst = np.asarray(Image.open(self.masks[ii]))
            cls = np.asarray(Image.open(self.categories[ii]))
            ids = np.unique(inst)
            ids = ids[ids != VOID_LABEL]
            n_obj = int(ids.max()) if ids.size else 0
            cats = []
            for obj in range(1, n_obj + 1):
                ys, xs = np.nonzero(inst == obj)
                cats.append(int(cls[ys[0], xs[0]]) if ys.size > self.area_thres else -1)
            self.obj_dict[im_id] = cats
        lines = ['\t"{:s}": {:s}'.format(k, json.dumps(self.obj_dict[k])) for k in self.im_ids]
        tmp = self.obj_list_file + f".tmp.{os.getpid()}"
        with open(tmp, "w") as fh:
            fh.write("{\n" + ",\n".join(lines) + "\n}\n")
        os.replace(tmp, self.obj_list_file)
        print("Preprocessing finished")

    # ------------------------------------------------------------------ archive handling
    def _check_integrity(self) -> bool:
        fpath = os.path.join(self.root, self.FILE)
        if not os.path.isfile(fpath):
            print("{} does not exist".format(fpath))
            return False
        md5 = hashlib.md5()
        with open(fpath, "rb") as fh:
            for block in iter(lambda: fh.read(1 << 20), b""):
                md5.update(block)
        if md5.hexdigest() != self.MD5:
            print(" MD5({}) did not match MD5({}) expected for {}".format(md5.hexdigest(), self.MD5, fpath))
            return False
        return True

    def _download(self) -> None:
        """Fetch + extract the archive (rank 0 only).  There is no network in the build/GPU
        environment, so this only succeeds where the URL is reachable."""
        fpath = os.path.join(self.root, self.FILE)
        with dist_env.master_first():
            if dist_env.is_master():
                os.makedirs(self.root, exist_ok=True)
                if self._check_integrity():
                    print("Files already downloaded and verified")
                else:
                    from urllib.request import urlretrieve
                    print("Downloading " + self.URL + " to " + fpath)
                    urlretrieve(self.URL, fpath)
                    print("Extracting tar file")
                    with tarfile.open(fpath) as tar:
                        tar.extractall(self.root)
                    print("Done!")


# --------------------------------------------------------------------------------------
# synthetic VOC-shaped data
# --------------------------------------------------------------------------------------
def _synthetic_scene(rng: np.random.Generator, h: int, w: int, max_objects: int = 4):
    """Random rectangles/ellipses: returns (rgb uint8, class map uint8 with 255 borders, instance map)."""
    cls = np.zeros((h, w), dtype=np.uint8)
    inst = np.zeros((h, w), dtype=np.uint8)
    img = rng.integers(0, 64, (h, w, 3), dtype=np.uint8)
    ys, xs = np.mgrid[0:h, 0:w]
    for k in range(1, int(rng.integers(1, max_objects + 1)) + 1):
        c = int(rng.integers(1, len(VOC_CLASSES)))
        cy, cx = rng.integers(h // 6, h - h // 6), rng.integers(w // 6, w - w // 6)
        ry, rx = rng.integers(h // 10, h // 3), rng.integers(w // 10, w // 3)
        if rng.random() < 0.5:
            region = ((ys - cy) / ry) ** 2 + ((xs - cx) / rx) ** 2 <= 1.0
        else:
            region = (np.abs(ys - cy) <= ry) & (np.abs(xs - cx) <= rx)
        ring = region & ~_erode(region)
        cls[region], inst[region] = c, k
        cls[ring], inst[ring] = VOID_LABEL, VOID_LABEL
        color = np.array([(37 * c) % 256, (91 * c + 60) % 256, (53 * c + 120) % 256], dtype=np.int32)
        noise = rng.integers(-20, 21, (int(region.sum()), 3))
        img[region] = np.clip(color[None, :] + noise, 0, 255).astype(np.uint8)
    return img, cls, inst


def _erode(m: np.ndarray) -> np.ndarray:
    p = np.pad(m, 1)
    return m & p[:-2, 1:-1] & p[2:, 1:-1] & p[1:-1, :-2] & p[1:-1, 2:]


class SyntheticVOC(data.Dataset):
    """Deterministic random scenes with VOC statistics.  ``mode='semantic'`` yields
    ``{'image' float32 HWC 0..255, 'gt' float32 HW class map with 255}``; ``mode='instance'`` yields
    the object-level dict of ``VOCSegmentation``."""

    category_names = list(VOC_CLASSES)

    def __init__(self, length: int = 256, size=(513, 513), transform: Optional[Callable] = None,
                 mode: str = "semantic", seed: int = 0, retname: bool = True):
        if mode not in ("semantic", "instance"):
            raise ValueError("mode must be 'semantic' or 'instance'")
        self.length, self.size, self.transform, self.mode, self.seed, self.retname = \
            length, tuple(size), transform, mode, seed, retname
        self.im_ids = [f"synthetic_{i:06d}" for i in range(length)]

    def __len__(self) -> int:
        return self.length

    def __getitem__(self, index: int) -> Dict:
        rng = np.random.default_rng(self.seed * 1_000_003 + index)
        img, cls, inst = _synthetic_scene(rng, *self.size)
        if self.mode == "semantic":
            sample = {"image": img.astype(np.float32), "gt": cls.astype(np.float32)}
        else:
            void = inst == VOID_LABEL
            target = (inst == 1) & ~void
            sample = {"image": img.astype(np.float32), "gt": target.astype(np.float32),
                      "void_pixels": void.astype(np.float32)}
        if self.retname:
            sample["meta"] = {"image": self.im_ids[index], "object": "0",
                              "category": int(cls[inst == 1][0]) if (inst == 1).any() else 0,
                              "im_size": self.size}
        return self.transform(sample) if self.transform is not None else sample

    def __str__(self) -> str:
        return f"SyntheticVOC(n={self.length},size={self.size},mode={self.mode})"


def write_synthetic_voc_tree(root: str, n_train: int = 6, n_val: int = 3, size=(96, 128), seed: int = 0) -> str:
    """Materialise a tiny VOC2012-layout tree (JPEG + palette-less PNGs + split files) under
    ``root`` — the fixture the dataset unit tests run against."""
    voc = os.path.join(root, VOCSegmentation.BASE_DIR)
    for d in ("JPEGImages", "SegmentationClass", "SegmentationObject", os.path.join("ImageSets", "Segmentation")):
        os.makedirs(os.path.join(voc, d), exist_ok=True)
    rng = np.random.default_rng(seed)
    ids = {"train": [], "val": []}
    for split, n in (("train", n_train), ("val", n_val)):
        for i in range(n):
            im_id = f"2099_{split}_{i:04d}"
            img, cls, inst = _synthetic_scene(rng, *size)
            Image.fromarray(img).save(os.path.join(voc, "JPEGImages", im_id + ".jpg"), quality=95)
            Image.fromarray(cls).save(os.path.join(voc, "SegmentationClass", im_id + ".png"))
            Image.fromarray(inst).save(os.path.join(voc, "SegmentationObject", im_id + ".png"))
            ids[split].append(im_id)
    for split, lst in ids.items():
        with open(os.path.join(voc, "ImageSets", "Segmentation", split + ".txt"), "w") as fh:
            fh.write("\n".join(lst) + "\n")
    return root
