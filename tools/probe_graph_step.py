#!/usr/bin/env python
"""Feasibility probe: does a whole training step (forward, fused loss, backward with the weight-gradient stream, fused
SGD, transposing refresh) capture into ONE hipGraph, and what does a replay cost against the eager step at small batch?
Learning rate and dropout keys are baked into the capture here (probe only).

    python tools/probe_graph_step.py [batch = 16] [model = deeplabv3plus_resnet101] [crop = 513]"""
import os
import sys
import time
from contextlib import contextmanager

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from distributedpytorch_amd.models import build_model  # noqa: E402
from distributedpytorch_amd.ops import SoftmaxCrossEntropy  # noqa: E402
from distributedpytorch_amd.parallel import TrainEngine  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
name = sys.argv[2] if len(sys.argv) > 2 else "deeplabv3plus_resnet101"
S = int(sys.argv[3]) if len(sys.argv) > 3 else 513
dev = torch.device("cuda", 0)
torch.manual_seed(0)
eng = TrainEngine(build_model(name, 21), SoftmaxCrossEntropy(255), lr=0.007, momentum=0.9, weight_decay=5e-4, device=dev)
x = torch.randn(B, 3, S, S, device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
y = torch.randint(0, 21, (B, S, S), device=dev, dtype=torch.uint8)


def timed(fn, n):
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t) / n


for _ in range(4):
    eng.train_step(x, y)                                  # tile choices are made here, outside any capture
print(f"eager: {timed(lambda: eng.train_step(x, y), 20):.2f} ms/step at B{B}", flush=True)


class _NoTimer:
    @contextmanager
    def phase(self, name):
        yield

    def summary(self, reset=True):
        return {}


eng.timer = _NoTimer()


def body():
    eng._step_events = []                                 # the host-side run-ahead bound syncs: not inside a capture
    return eng.train_step(x, y)


side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    for _ in range(2):
        body()
torch.cuda.current_stream().wait_stream(side)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
try:
    with torch.cuda.graph(g):
        loss = body()
except Exception as e:                                    # noqa: BLE001
    print("CAPTURE FAILED:", type(e).__name__, str(e)[:600], flush=True)
    sys.exit(2)
torch.cuda.synchronize()
print("captured", flush=True)
g.replay()
torch.cuda.synchronize()
print(f"replay: {timed(g.replay, 20):.2f} ms/step, loss {float(loss):.4f}", flush=True)
