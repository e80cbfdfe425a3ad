#!/usr/bin/env python
"""Host time of one GpuBatchLoader batch with an EMPTY device queue (a synchronisation before every fetch), next to the
device time of its kernels: separates the loader's own host cost from back-pressure of a full launch queue — the
`data_wait_ms` of the training log is host wall time inside the loader, whichever of the two it is.

    python tools/loader_host_cost.py [--task instance|semseg] [--batch 64]
"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import train_pascal as tp  # noqa: E402
from distributedpytorch_amd.data.gpu_pipeline import GpuBatchLoader  # noqa: E402
from distributedpytorch_amd.parallel.sampler import DistributedSampler  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--task", default="instance")
ap.add_argument("--batch", type=int, default=64)
ap.add_argument("--len", type=int, default=1024)
a = ap.parse_args()
instance = a.task == "instance"
args = tp.parse_args(["--synthetic", "--task", a.task, "--synthetic-len", str(a.len)])
args.crop = 512 if instance else 513
args.nInputChannels = 4 if instance else 3
dev = torch.device("cuda")
db_train, _ = tp.build_datasets(args, raw=True)
sampler = DistributedSampler(len(db_train), shuffle=True, seed=0, drop_last=True)
rots, scales = ((-20, 20), (.75, 1.25)) if instance else ((0.0, 0.0), (0.5, 2.0))
loader = GpuBatchLoader(db_train, sampler, a.batch, dev, args.crop, a.task, scales=scales, rots=rots, seed=0, rank=0,
                        workers=12, hbm_budget_gb=64.0, relax=50, zero_pad=True)
loader.prepare()
it = iter(loader)
host, total = [], []
for i in range(12):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    x, y = next(it)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    host.append(1e3 * (t1 - t0))
    total.append(1e3 * (t2 - t0))
print(f"{a.task} batch {a.batch}: host ms per batch (empty queue) {[round(v, 2) for v in host]}")
print(f"{a.task} batch {a.batch}: host + device ms per batch {[round(v, 2) for v in total]}")
