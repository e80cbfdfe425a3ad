#!/usr/bin/env python
"""Training steps and hipGraph-replayed evaluation interleaved in one process: the graph is captured once (eval mode)
and must see the weights and running statistics of every later step; the replayed metrics equal the eager ones."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from distributedpytorch_amd.models import build_model  # noqa: E402
from distributedpytorch_amd.ops import SoftmaxCrossEntropy  # noqa: E402
from distributedpytorch_amd.parallel import TrainEngine  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "deeplabv3plus_resnet50"
dev = torch.device("cuda", 0)
torch.manual_seed(0)
eng = TrainEngine(build_model(name, 21), SoftmaxCrossEntropy(255), lr=0.01, momentum=0.9, device=dev)
g = torch.Generator().manual_seed(1)
train = [(torch.randn(8, 3, 129, 129, generator=g), torch.randint(0, 21, (8, 129, 129), generator=g, dtype=torch.uint8))
         for _ in range(6)]
val = [(torch.randn(1, 3, 129, 129, generator=g), torch.randint(0, 21, (1, 129, 129), generator=g, dtype=torch.uint8))
       for _ in range(6)]
ok = True
for rnd in range(3):
    for x, y in train[2 * rnd:2 * rnd + 2]:
        loss = eng.train_step(x, y)
    a = eng.evaluate(val, 21, graph=True)
    b = eng.evaluate(val, 21, graph=False)
    same = abs(a["miou"] - b["miou"]) < 1e-6 and abs(a["loss"] - b["loss"]) < 1e-4 * max(1.0, abs(b["loss"]))
    ok &= same
    print(f"round {rnd}: train loss {float(loss):.4f}  graph mIoU {a['miou']:.6f} loss {a['loss']:.5f} | eager mIoU "
          f"{b['miou']:.6f} loss {b['loss']:.5f} | {'same' if same else 'DIFFERENT'}", flush=True)
print("ok" if ok else "MISMATCH")
sys.exit(0 if ok else 1)
