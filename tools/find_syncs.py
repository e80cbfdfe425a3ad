#!/usr/bin/env python
"""Which Python lines synchronise host and device in a steady-state training iteration?

Runs `train_pascal.py` (arguments after `--`) and, from the 4th training step on, switches PyTorch's synchronisation
debug mode to "warn": every blocking copy / `.item()` / `nonzero` on the device then reports the line that issued it.
Found with it: nothing in the semseg step; the instance loader's per-batch blocking copy (profiles/r3_instance_loader_sync.md).

    python tools/find_syncs.py -- --synthetic --task instance --trainBatch 32 --max-iters 8 --nEpochs 1 --useTest 0 --log-every 0
"""
import collections
import os
import runpy
import sys
import traceback
import warnings

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from distributedpytorch_amd.parallel import engine as E  # noqa: E402

seen = collections.Counter()
calls = [0]
orig_step = E.TrainEngine.train_step


def show(message, category, filename, lineno, file=None, line=None):
    if "synchroniz" not in str(message):
        return
    stack = [f for f in traceback.extract_stack()[:-1] if ROOT in f.filename and "find_syncs" not in f.filename]
    where = " <- ".join(f"{os.path.relpath(f.filename, ROOT)}:{f.lineno}" for f in reversed(stack[-4:]))
    seen[where] += 1


def step(self, *a, **k):
    calls[0] += 1
    if calls[0] == 4:
        warnings.simplefilter("always")
        warnings.showwarning = show
        torch.cuda.set_sync_debug_mode("warn")
    return orig_step(self, *a, **k)


E.TrainEngine.train_step = step
argv = sys.argv[sys.argv.index("--") + 1:] if "--" in sys.argv else []
sys.argv = ["train_pascal.py"] + argv
try:
    runpy.run_path(os.path.join(ROOT, "train_pascal.py"), run_name="__main__")
finally:
    torch.cuda.set_sync_debug_mode("default")
    n = max(calls[0] - 3, 1)
    print(f"synchronising calls after step 3 ({n} steps observed, including end-of-epoch work):")
    for where, c in seen.most_common():
        print(f"  {c:4d} x  {where}")
    if not seen:
        print("  none")
