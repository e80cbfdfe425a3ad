#!/usr/bin/env python
"""BatchNorm kernel micro-benchmark: achieved GB/s of apply / bwd_reduce / bwd_apply per
representative DeepLabv3+ R101 activation shape (bytes = minimum traffic of the pass)."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from distributedpytorch_amd.ops import _hip_bn  # noqa: E402

SHAPES = [  # (name, count, C, HW, residual)
    ("stem", 1, 64, 257, False), ("l1.64", 6, 64, 129, False), ("l1.256", 4, 256, 129, True),
    ("l2.128", 8, 128, 65, False), ("l2.512", 5, 512, 65, True),
    ("l3.256", 46, 256, 33, False), ("l3.1024", 24, 1024, 33, True),
    ("l4.512", 6, 512, 33, False), ("l4.2048", 4, 2048, 33, True),
    ("aspp.256", 6, 256, 33, False), ("dec.256", 2, 256, 129, False), ("dec.48", 1, 48, 129, False),
]


def timeit(fn, iters=7, warm=2):
    for _ in range(warm):
        fn()
    ts = []
    for _ in range(iters):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); b.synchronize()
        ts.append(a.elapsed_time(b) * 1e3)
    return sorted(ts)[len(ts) // 2]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=32)
    args = ap.parse_args()
    dev = "cuda"
    tot = [0.0, 0.0, 0.0]
    print(f"{'shape':10s} {'cnt':>3s} {'MB':>7s} | {'apply us':>8s} {'GB/s':>6s} | {'reduce us':>9s} {'GB/s':>6s} | {'bapply us':>9s} {'GB/s':>6s}")
    for name, cnt, c, hw, res in SHAPES:
        n = args.batch
        x = torch.randn(n, c, hw, hw, device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
        dy = torch.randn_like(x)
        r = torch.randn_like(x) if res else None
        mb = x.numel() * 2 / 1e6
        w, b = torch.ones(c, device=dev), torch.zeros(c, device=dev)
        rm, rv = torch.zeros(c, device=dev), torch.ones(c, device=dev)
        s, ss = _hip_bn.stats(x)
        cntf = float(x.numel() // c)
        y, mean, invstd = _hip_bn.apply_from_stats(x, s, ss, w, b, r, rm, rv, cntf, 1e-5, 0.1, True)
        mode = 1 if res else 2
        sdy, sdyx = _hip_bn.bwd_reduce(dy, x, y, mean, invstd, w, b, mode)
        sdy, sdyx = sdy.clone(), sdyx.clone()
        t_a = timeit(lambda: _hip_bn.apply_from_stats(x, s, ss, w, b, r, rm, rv, cntf, 1e-5, 0.1, True))
        t_r = timeit(lambda: _hip_bn.bwd_reduce(dy, x, y, mean, invstd, w, b, mode))
        t_b = timeit(lambda: _hip_bn.bwd_apply(dy, x, y, mean, invstd, w, b, sdy, sdyx, cntf, mode, res))
        na, nr, nb = (3 if res else 2), (3 if res else 2), (5 if res else 3)
        print(f"{name:10s} {cnt:3d} {mb:7.1f} | {t_a:8.1f} {na * mb / t_a * 1e3:6.0f} | {t_r:9.1f} {nr * mb / t_r * 1e3:6.0f} | "
              f"{t_b:9.1f} {nb * mb / t_b * 1e3:6.0f}", flush=True)
        tot[0] += cnt * t_a; tot[1] += cnt * t_r; tot[2] += cnt * t_b
    print(f"TOTAL weighted (ms/step): apply {tot[0] / 1e3:.2f}  bwd_reduce {tot[1] / 1e3:.2f}  bwd_apply {tot[2] / 1e3:.2f}")


if __name__ == "__main__":
    main()
