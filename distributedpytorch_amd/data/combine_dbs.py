"""``dataloaders.combine_dbs.CombineDBs`` (`/root/reference/train_pascal.py:27`, `:152`): the union
of several object-level datasets minus every image that appears in an ``excluded`` dataset
(e.g. train on VOC-train + SBD but never on an image of VOC-val).  Images present in more than one
included dataset are taken from the first one only."""
from __future__ import annotations

from typing import List, Optional, Sequence

from torch.utils.data import Dataset


def _image_id_of(db, index: int) -> str:
    if hasattr(db, "obj_list") and hasattr(db, "im_ids"):
        return str(db.im_ids[db.obj_list[index][0]])
    if hasattr(db, "im_ids"):
        return str(db.im_ids[index])
    return f"{id(db)}:{index}"


class CombineDBs(Dataset):
    def __init__(self, dataloaders: Sequence, excluded: Optional[Sequence] = None):
        self.dataloaders = list(dataloaders)
        self.excluded = list(excluded or [])
        banned = set()
        for db in self.excluded:
            banned.update(str(i) for i in getattr(db, "im_ids", []))
        self.cat_list: List[dict] = []
        self.im_ids: List[str] = []
        owner = {}
        n_excluded = 0
        for di, db in enumerate(self.dataloaders):
            for idx in range(len(db)):
                im = _image_id_of(db, idx)
                if im in banned:
                    n_excluded += 1
                    continue
                if owner.setdefault(im, di) != di:
                    continue  # same image already provided by an earlier dataset
                self.cat_list.append({"db_ii": di, "cat_ii": idx})
                if not self.im_ids or self.im_ids[-1] != im:
                    self.im_ids.append(im)
        self.n_excluded = n_excluded
        print(f"Combined number of images: {len(set(self.im_ids)):d}\nCombined number of objects: {len(self.cat_list):d}")

    def __getitem__(self, index):
        ent = self.cat_list[index]
        return self.dataloaders[ent["db_ii"]][ent["cat_ii"]]

    def __len__(self):
        return len(self.cat_list)

    def __str__(self):
        inc = "+".join(str(db) for db in self.dataloaders)
        exc = "+".join(str(db) for db in self.excluded)
        return f"Included datasets:{inc}\nExcluded datasets:{exc}"
