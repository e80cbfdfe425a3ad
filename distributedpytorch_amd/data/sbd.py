"""Semantic Boundaries Dataset (SBD) — the second source the reference combines with VOC when
``use_sbd`` is on: ``CombineDBs([voc_train, sbd], excluded=[voc_val])``
(`/root/reference/train_pascal.py:29`, `:150-152`; the ``sbd`` module itself is not in the reference
tree and its import is commented out, so that branch would raise ``NameError`` there).

Layout read here (the public ``benchmark_RELEASE`` archive)::

    <root>/benchmark_RELEASE/dataset/{img/<id>.jpg, cls/<id>.mat, inst/<id>.mat, train.txt, val.txt}

``cls/<id>.mat`` holds ``GTcls.Segmentation`` (class map, 0 = background), ``inst/<id>.mat`` holds
``GTinst.Segmentation`` (instance ids) and ``GTinst.Categories``.  Samples follow the dict contract of
``VOCSegmentation`` (`/root/reference/pascal.py:122-137`): per-object ``{'image','gt','void_pixels'}``
(+ ``'meta'``) or, with ``semantic=True``, per-image ``{'image','gt'}`` class maps (SBD has no void
label).  ``.mat`` files are parsed with ``scipy.io.loadmat`` (plain MATLAB v5 arrays, nothing is
unpickled)."""
from __future__ import annotations

import os
from typing import Callable, Dict, List, Optional, Sequence, Union

import numpy as np
import torch.utils.data as data
from PIL import Image

from .pascal import VOC_CLASSES


def _loadmat(path: str, key: str):
    from scipy.io import loadmat
    m = loadmat(path, struct_as_record=False, squeeze_me=True)
    return m[key]


class SBDSegmentation(data.Dataset):
    BASE_DIR = os.path.join("benchmark_RELEASE", "dataset")
    category_names = list(VOC_CLASSES)

    def __init__(self, root: str, split: Union[str, Sequence[str]] = "val", transform: Optional[Callable] = None,
                 area_thres: int = 0, retname: bool = True, semantic: bool = False):
        self.root, self.transform, self.area_thres, self.retname, self.semantic = \
            root, transform, area_thres, retname, semantic
        self.split = [split] if isinstance(split, str) else sorted(split)
        base = os.path.join(root, self.BASE_DIR)
        if not os.path.isdir(base):
            base = root                                   # also accept the dataset directory itself
        self._img, self._cls, self._inst = (os.path.join(base, d) for d in ("img", "cls", "inst"))
        self.im_ids: List[str] = []
        for name in self.split:
            with open(os.path.join(base, name + ".txt")) as fh:
                for ln in fh.read().splitlines():
                    if ln.strip() and ln.strip() not in self.im_ids:
                        self.im_ids.append(ln.strip())
        for im_id in self.im_ids:
            for p in (os.path.join(self._img, im_id + ".jpg"), os.path.join(self._cls, im_id + ".mat"),
                      os.path.join(self._inst, im_id + ".mat")):
                if not os.path.isfile(p):
                    raise FileNotFoundError(p)
        self.obj_list: List[List[int]] = []
        self._cats: Dict[str, List[int]] = {}
        if not semantic:
            for ii, im_id in enumerate(self.im_ids):
                gt = _loadmat(os.path.join(self._inst, im_id + ".mat"), "GTinst")
                seg = np.asarray(gt.Segmentation)
                cats = np.atleast_1d(np.asarray(gt.Categories)).astype(int).tolist()
                keep = []
                for jj, c in enumerate(cats):
                    area = int((seg == jj + 1).sum())
                    keep.append(int(c) if area > area_thres else -1)
                    if keep[-1] != -1:
                        self.obj_list.append([ii, jj])
                self._cats[im_id] = keep

    def __len__(self) -> int:
        return len(self.im_ids) if self.semantic else len(self.obj_list)

    def __str__(self) -> str:
        return "SBD(split=" + str(self.split) + ",area_thres=" + str(self.area_thres) + ")"

    def _rgb(self, im_id: str) -> np.ndarray:
        return np.asarray(Image.open(os.path.join(self._img, im_id + ".jpg")).convert("RGB"), dtype=np.float32)

    def __getitem__(self, index: int) -> Dict:
        if self.semantic:
            im_id = self.im_ids[index]
            cls = np.asarray(_loadmat(os.path.join(self._cls, im_id + ".mat"), "GTcls").Segmentation, dtype=np.float32)
            sample = {"image": self._rgb(im_id), "gt": cls}
            if self.retname:
                sample["meta"] = {"image": im_id, "im_size": cls.shape[:2]}
        else:
            ii, jj = self.obj_list[index]
            im_id = self.im_ids[ii]
            seg = np.asarray(_loadmat(os.path.join(self._inst, im_id + ".mat"), "GTinst").Segmentation)
            gt = (seg == jj + 1).astype(np.float32)
            sample = {"image": self._rgb(im_id), "gt": gt, "void_pixels": np.zeros_like(gt)}
            if self.retname:
                sample["meta"] = {"image": im_id, "object": str(jj), "category": self._cats[im_id][jj],
                                  "im_size": gt.shape[:2]}
        return self.transform(sample) if self.transform is not None else sample


def write_synthetic_sbd_tree(root: str, n_train: int = 4, n_val: int = 2, size=(64, 80), seed: int = 0,
                             shared_ids: Sequence[str] = ()) -> str:
    """Tiny SBD-layout fixture (``.mat`` written with ``scipy.io.savemat``) for the unit tests; ``shared_ids`` are
    image ids that also exist in a VOC fixture (to exercise ``CombineDBs``' exclusion)."""
    from scipy.io import savemat

    from .pascal import _synthetic_scene
    base = os.path.join(root, SBDSegmentation.BASE_DIR)
    for d in ("img", "cls", "inst"):
        os.makedirs(os.path.join(base, d), exist_ok=True)
    rng = np.random.default_rng(seed)
    ids = {"train": [], "val": []}
    extra = list(shared_ids)
    for split, n in (("train", n_train), ("val", n_val)):
        for i in range(n):
            im_id = extra.pop() if (extra and split == "train") else f"2098_{split}_{i:04d}"
            img, cls, inst = _synthetic_scene(rng, *size)
            cls, inst = cls.copy(), inst.copy()
            cls[cls == 255], inst[inst == 255] = 0, 0
            n_obj = int(inst.max())
            cats = [int(cls[inst == k + 1][0]) if (inst == k + 1).any() else 0 for k in range(n_obj)]
            Image.fromarray(img).save(os.path.join(base, "img", im_id + ".jpg"), quality=95)
            savemat(os.path.join(base, "cls", im_id + ".mat"), {"GTcls": {"Segmentation": cls}})
            savemat(os.path.join(base, "inst", im_id + ".mat"),
                    {"GTinst": {"Segmentation": inst, "Categories": np.asarray(cats, dtype=np.uint8).reshape(-1, 1)}})
            ids[split].append(im_id)
    for split, lst in ids.items():
        with open(os.path.join(base, split + ".txt"), "w") as fh:
            fh.write("\n".join(lst) + "\n")
    return root
