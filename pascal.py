"""Top-level ``pascal`` module (same name as `/root/reference/pascal.py`): the implementation
lives in ``distributedpytorch_amd.data.pascal``."""
from distributedpytorch_amd.data.pascal import (VOC_CLASSES, VOID_LABEL, SyntheticVOC,  # noqa: F401
                                                VOCSegmentation, write_synthetic_voc_tree)

if __name__ == "__main__":
    import sys

    import numpy as np

    from distributedpytorch_amd.data import helpers
    from distributedpytorch_amd.data import transforms as tr

    # smoke run mirroring the reference's __main__ (no display needed: writes PNG overlays)
    root = sys.argv[1] if len(sys.argv) > 1 else None
    if root is None:
        ds = SyntheticVOC(length=4, size=(128, 160), mode="instance", transform=tr.Compose([tr.ToTensor()]))
    else:
        ds = VOCSegmentation(root=root, split=["train", "val"], transform=tr.Compose([tr.ToTensor()]),
                             retname=True, check_integrity=False)
    from PIL import Image
    for i in range(min(4, len(ds))):
        s = ds[i]
        ov = helpers.overlay_mask(helpers.tens2image(s["image"]) / 255., np.squeeze(helpers.tens2image(s["gt"])))
        Image.fromarray((ov * 255).astype(np.uint8)).save(f"pascal_overlay_{i}.png")
        print(i, ds.category_names[s["meta"]["category"]])
