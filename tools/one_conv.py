#!/usr/bin/env python
"""Run ONE conv shape a few times (for rocprofv3 --pmc passes on a single kernel).

    python tools/one_conv.py --cin 1024 --cout 256 --hw 33 --k 1 [--dil 1] [--stride 1] [--batch 32]
        [--pass fwd|dgrad|wgrad] [--tile 0] [--iters 5]
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from distributedpytorch_amd.ops import _hip_conv  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cin", type=int, default=1024)
    ap.add_argument("--cout", type=int, default=256)
    ap.add_argument("--hw", type=int, default=33)
    ap.add_argument("--k", type=int, default=1)
    ap.add_argument("--dil", type=int, default=1)
    ap.add_argument("--stride", type=int, default=1)
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--pass", dest="which", default="fwd")
    ap.add_argument("--tile", type=int, default=0)
    ap.add_argument("--iters", type=int, default=5)
    ap.add_argument("--splits", type=int, default=0, help="wgrad split-K count (0 = cost model / autotune)")
    a = ap.parse_args()
    pad = a.dil * (a.k - 1) // 2
    ho = (a.hw + 2 * pad - a.dil * (a.k - 1) - 1) // a.stride + 1
    dev = "cuda"
    x = torch.randn(a.batch, a.cin, a.hw, a.hw, device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    w = (torch.randn(a.cout, a.cin, a.k, a.k, device=dev) / (a.cin * a.k * a.k) ** 0.5).to(torch.bfloat16) \
        .contiguous(memory_format=torch.channels_last)
    dy = torch.randn(a.batch, a.cout, ho, ho, device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    wt = _hip_conv.weight_transpose(w)
    pad_t = a.dil * (a.k - 1) - pad
    for _ in range(a.iters):
        if a.which == "fwd":
            _hip_conv.igemm(x, w, (ho, ho), a.stride, 1, (pad, pad), (a.dil, a.dil), tile=a.tile)
        elif a.which == "dgrad":
            _hip_conv.igemm(dy, wt, (a.hw, a.hw), 1, a.stride, (pad_t, pad_t), (a.dil, a.dil), tile=a.tile)
        else:
            _hip_conv.wgrad(dy, x, (a.cout, a.cin, a.k, a.k), (a.stride, a.stride), (pad, pad), (a.dil, a.dil),
                            splits=a.splits)
    torch.cuda.synchronize()
    print("ok")


if __name__ == "__main__":
    main()
