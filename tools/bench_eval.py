#!/usr/bin/env python
"""Batch-1 inference latency, eager launches vs hipGraph replay (parallel.graphs)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from distributedpytorch_amd.models import build_model  # noqa: E402
from distributedpytorch_amd.parallel import GraphedForward  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "deeplabv3plus_resnet101"
bs = int(sys.argv[2]) if len(sys.argv) > 2 else 1
model = build_model(name, 21).cuda().to(memory_format=torch.channels_last).eval()
x = torch.randn(bs, 3, 513, 513, device="cuda").to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
g = GraphedForward(model)


def timeit(fn, n=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


with torch.no_grad():
    te = timeit(lambda: model(x))
tg = timeit(lambda: g(x))
print(f"{name} batch {bs} 513x513 inference: eager {te:.3f} ms, hipGraph {tg:.3f} ms ({te / tg:.2f}x)")
