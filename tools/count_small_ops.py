#!/usr/bin/env python
"""Which ATen ops (and how many per step) does a native training step still launch?  torch.profiler over 2 steps of
DeepLabv3+ R101 at a small batch; prints op counts per step and the Python call sites of the small device copies."""
import os
import sys

import numpy as np
import torch
from torch.profiler import ProfilerActivity, profile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from distributedpytorch_amd.models import build_model  # noqa: E402
from distributedpytorch_amd.ops import SoftmaxCrossEntropy  # noqa: E402
from distributedpytorch_amd.parallel import TrainEngine  # noqa: E402


def main():
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    name = sys.argv[1] if len(sys.argv) > 1 else "deeplabv3plus_resnet101"
    os8 = "--os8" in sys.argv
    B, S = 8, 257
    if name.startswith("danet"):             # the reference's own workload: 4-channel input, binary mask, 3 heads
        from distributedpytorch_amd.utils import SegmentationMultiLosses
        model = build_model(name, 1, in_channels=4)
        eng = TrainEngine(model, SegmentationMultiLosses(1), lr=5e-8, momentum=0.9, weight_decay=5e-4, device=dev)
        S = 256
        x = (torch.rand(B, 4, S, S, device=dev) * 255).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
        y = (torch.rand(B, 1, S, S, device=dev) > 0.5).float()
    else:
        model = build_model(name, 21, **({"output_stride": 8} if os8 else {}))
        eng = TrainEngine(model, SoftmaxCrossEntropy(255), lr=0.007, momentum=0.9, weight_decay=5e-4, device=dev)
        x = torch.randn(B, 3, S, S, device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
        y = torch.randint(0, 21, (B, S, S), device=dev).to(torch.uint8)
    for _ in range(3):
        eng.train_step(x, y)
    torch.cuda.synchronize()
    steps = 2
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
        for _ in range(steps):
            eng.train_step(x, y)
        torch.cuda.synchronize()
    rows = [(e.key, e.count / steps, e.device_time_total / steps) for e in prof.key_averages() if e.count >= steps]
    rows.sort(key=lambda r: -r[1])
    print(f"{'op / kernel':70s} {'per step':>9s} {'device us/step':>15s}")
    for k, c, t in rows[:45]:
        print(f"{k[:70]:70s} {c:9.1f} {t:15.1f}")
    print("\ncall sites of aten::copy_ / aten::fill_ / aten::zero_ / aten::add_ / aten::sum (top stacks):")
    for e in prof.key_averages(group_by_stack_n=6):
        if e.key in ("aten::copy_", "aten::fill_", "aten::zero_", "aten::add_", "aten::sum", "aten::mul", "aten::to", "aten::_to_copy",
                     "aten::add", "aten::mm", "aten::zeros", "aten::empty", "aten::cat", "aten::div", "aten::sub") and e.count >= 4 * steps:
            st = [s for s in e.stack if "distributedpytorch_amd" in s or "train" in s][:3]
            print(f"  {e.key:14s} x{e.count / steps:6.1f}/step  {' <- '.join(s.split('/')[-1] for s in st)}")


if __name__ == "__main__":
    main()
