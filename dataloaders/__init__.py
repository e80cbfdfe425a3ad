"""``dataloaders`` package: the import paths of the reference script resolve here
(`/root/reference/train_pascal.py:27-36`: ``dataloaders.pascal``, ``dataloaders.custom_transforms``,
``dataloaders.helpers``, ``dataloaders.implementation``, ``dataloaders.combine_dbs``, plus
``dataloaders.nellipse`` / ``dataloaders.skewed_axes_weight_map`` from `custom_transforms.py:6-7`)."""
