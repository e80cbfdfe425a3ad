#!/usr/bin/env python
"""Time forward / dgrad / wgrad of a list of conv shapes (median of N runs, autotuned tiles) — a small companion of
tools/bench_conv.py for ad-hoc shapes:  python tools/bench_shapes.py --batch 64 728:728:65:1 256:1024:33:1 ..."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from distributedpytorch_amd.ops import _hip_conv  # noqa: E402


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
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--cfgs", default="", help="comma-separated tile configurations to time the forward with (else autotuned)")
    ap.add_argument("shapes", nargs="+", help="cin:cout:hw:k[:dil]")
    a = ap.parse_args()
    print(f"{'shape':22s} {'GF':>8s} | {'fwd us':>8s} {'TF':>6s} | {'dgrad us':>8s} {'TF':>6s} | {'wgrad us':>8s} {'TF':>6s}")
    for sh in a.shapes:
        f = [int(v) for v in sh.split(":")]
        cin, cout, hw, k = f[:4]
        dil = f[4] if len(f) > 4 else 1
        pad = dil * (k - 1) // 2
        x = torch.randn(a.batch, cin, hw, hw, device="cuda").to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
        w = (torch.randn(cout, cin, k, k, device="cuda") / (cin * k * k) ** 0.5).to(torch.bfloat16) \
            .contiguous(memory_format=torch.channels_last)
        dy = torch.randn(a.batch, cout, hw, hw, device="cuda").to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
        wt = _hip_conv.weight_transpose(w)
        dw = torch.zeros(cout, cin, k, k, device="cuda").contiguous(memory_format=torch.channels_last)
        gf = 2.0 * a.batch * hw * hw * cin * cout * k * k / 1e9
        for c in [int(v) for v in a.cfgs.split(",") if v]:
            t_c = timeit(lambda: _hip_conv.igemm(x, w, (hw, hw), 1, 1, (pad, pad), (dil, dil), cfg=c))
            print(f"    cfg {c:2d}: {t_c:8.1f} us {gf / t_c * 1e3:6.0f} TF", flush=True)
        t_f = timeit(lambda: _hip_conv.igemm(x, w, (hw, hw), 1, 1, (pad, pad), (dil, dil)))
        t_d = timeit(lambda: _hip_conv.igemm(dy, wt, (hw, hw), 1, 1, (pad, pad), (dil, dil)))
        t_w = timeit(lambda: _hip_conv.wgrad(dy, x, (cout, cin, k, k), (1, 1), (pad, pad), (dil, dil), out=dw))
        print(f"{sh:22s} {gf:8.1f} | {t_f:8.1f} {gf / t_f * 1e3:6.0f} | {t_d:8.1f} {gf / t_d * 1e3:6.0f} | {t_w:8.1f} {gf / t_w * 1e3:6.0f}",
              flush=True)


if __name__ == "__main__":
    main()
