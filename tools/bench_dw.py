#!/usr/bin/env python
"""Depthwise 3x3 micro-benchmark (csrc/depthwise.hip) on the Xception-65 OS8 shapes.

    python tools/bench_dw.py [--batch 16]
Prints us per call and the effective HBM rate of the ideal traffic (fwd / dgrad: read + write one
tensor; wgrad: read two)."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from distributedpytorch_amd import ops  # noqa: E402

SHAPES = [  # name, count, C, HW, dil
    ("entry.128@129", 2, 128, 129, 1),
    ("entry.256@65", 2, 256, 65, 1),
    ("middle.728@65.d2", 0, 728, 65, 2),
    ("middle.768@65.d2", 51, 768, 65, 2),      # the 728-channel flow as it is stored (models/xception.py)
    ("exit.1024@65.d2", 1, 1024, 65, 2),
    ("exit.1536@65.d4", 2, 1536, 65, 4),
]


def timeit(fn, iters=5, inner=10):
    fn()
    ts = []
    for _ in range(iters):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        a.record()
        for _ in range(inner):
            fn()
        b.record()
        b.synchronize()
        ts.append(a.elapsed_time(b) * 1e3 / inner)
    return sorted(ts)[len(ts) // 2]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--flat", action="store_true", help="the flat kernels instead of the rolling-window ones")
    args = ap.parse_args()
    from distributedpytorch_amd.ops import _hip_conv
    if args.flat:
        _hip_conv._lib().dp_dwconv_set_window(0)
    tot = [0.0, 0.0, 0.0]
    for name, cnt, c, hw, d in SHAPES:
        x = torch.randn(args.batch, c, hw, hw, device="cuda").to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
        w = torch.randn(c, 1, 3, 3, device="cuda")
        w16 = w.to(torch.bfloat16)
        xg = x.clone().requires_grad_(True)
        wg = w.clone().requires_grad_(True)
        y = _hip_conv._DepthwiseConv.apply(xg, wg, w16, (1, 1), (d, d), (d, d))
        dy = torch.randn_like(y)
        from distributedpytorch_amd.ops import _backend as B
        lib = _hip_conv._lib()
        w9 = w16.reshape(c, 9).t().contiguous()
        dx = torch.empty_like(x)
        dw9 = torch.zeros(9, c, device="cuda")
        hw_ = hw
        # the launchers themselves, on the current stream (through autograd the weight gradient runs on the auxiliary
        # stream and the event pair would not see it)
        t_f = timeit(lambda: lib.dp_dwconv_fwd(B.ptr(x), B.ptr(w9), B.ptr(dx), args.batch, c, hw_, hw_, hw_, hw_, 1, d, d, d,
                                               B.stream_ptr()))
        t_dx = timeit(lambda: lib.dp_dwconv_bwd_data(B.ptr(dy), B.ptr(w9), B.ptr(dx), args.batch, c, hw_, hw_, hw_, hw_, 1,
                                                     d, d, d, B.stream_ptr()))
        t_w = timeit(lambda: lib.dp_dwconv_bwd_weight(B.ptr(dy), B.ptr(x), B.ptr(dw9), 1, args.batch, c, hw_, hw_, hw_, hw_, 1,
                                                      d, d, d, B.stream_ptr()))
        mb = x.numel() * 2 / 1e6
        print(f"{name:18s} x{cnt:2d} {mb:7.1f} MB | fwd {t_f:7.1f} us ({2 * mb / t_f:5.2f} TB/s) | dgrad {t_dx:7.1f} us "
              f"({2 * mb / t_dx:5.2f} TB/s) | wgrad {t_w:7.1f} us ({2 * mb / t_w:5.2f} TB/s)", flush=True)
        for i, t in enumerate((t_f, t_dx, t_w)):
            tot[i] += cnt * t
    print(f"TOTAL ms/step: fwd {tot[0] / 1e3:.2f}  dgrad {tot[1] / 1e3:.2f}  wgrad {tot[2] / 1e3:.2f}")


if __name__ == "__main__":
    main()
