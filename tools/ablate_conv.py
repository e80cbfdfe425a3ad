#!/usr/bin/env python
"""Which resource bounds a conv_igemm2 tile config?  Times one layer with parts of the K loop removed
(csrc/conv_igemm2.hip, Conv2Args::ablate): full / no LDS-DMA in the loop / no MFMA / DMA only.

    python tools/ablate_conv.py --cin 2048 --cout 256 --hw 33 --k 3 --dil 12 --batch 128 --cfgs 5,9
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from distributedpytorch_amd.ops import _backend as B  # noqa: E402
from distributedpytorch_amd.ops import _hip_conv  # noqa: E402


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
    ap.add_argument("--cin", type=int, default=2048)
    ap.add_argument("--cout", type=int, default=256)
    ap.add_argument("--hw", type=int, default=33)
    ap.add_argument("--k", type=int, default=3)
    ap.add_argument("--dil", type=int, default=12)
    ap.add_argument("--batch", type=int, default=128)
    ap.add_argument("--cfgs", default="5,9")
    a = ap.parse_args()
    pad = a.dil * (a.k - 1) // 2
    x = torch.randn(a.batch, a.cin, a.hw, a.hw, device="cuda").to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    w = (torch.randn(a.cout, a.cin, a.k, a.k, device="cuda") / (a.cin * a.k * a.k) ** 0.5).to(torch.bfloat16) \
        .contiguous(memory_format=torch.channels_last)
    gf = 2.0 * a.batch * a.hw * a.hw * a.cout * a.cin * a.k * a.k / 1e9
    lib = B.lib()
    names = {0: "full", 1: "no DMA in loop", 2: "no MFMA", 3: "reads only", 4: "DMA only", 5: "barriers only"}
    for cfg in [int(c) for c in a.cfgs.split(",")]:
        for mask in (0, 1, 2, 3, 4, 5):
            lib.dp_conv_set_ablate(mask)
            t = timeit(lambda: _hip_conv.igemm(x, w, (a.hw, a.hw), 1, 1, (pad, pad), (a.dil, a.dil), cfg=cfg))
            print(f"cfg {cfg:2d} {names[mask]:16s} {t:8.1f} us  ({gf / t * 1e3:7.1f} TF/s equivalent)", flush=True)
    lib.dp_conv_set_ablate(0)


if __name__ == "__main__":
    main()
