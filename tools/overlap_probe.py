"""Can a memory-bound BatchNorm pass and an MFMA-bound weight-gradient kernel share the chip?

The R101 step's time equals the SUM of its kernels' times alone (README "Where the time goes"): the weight gradients
on the auxiliary stream cost the step 26 of their 29 ms.  This probe times, on the layer-3 bottleneck shapes,

    A  = bn_bwd_apply over a [M, 1024] tensor           (HBM-bound, no LDS)
    B  = weight gradient of the 1x1 256 -> 1024 conv     (LDS-DMA + MFMA, 144 KiB LDS, one block per CU)
    C  = data gradient of the 3x3 256 -> 256 conv        (MFMA-bound)

alone, and A || B, C || B, A || C on two streams (the second one at the device's lowest priority, as ops.streams
creates it, or at normal priority), so that the schedule of backward can be built on what the hardware can overlap.

    python tools/overlap_probe.py [--batch 128]
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from distributedpytorch_amd.ops import _hip_bn, _hip_conv, streams  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=128)
    ap.add_argument("--iters", type=int, default=20)
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    n, hw = args.batch, 33
    cl = torch.channels_last

    def t(c):
        return torch.randn(n, c, hw, hw, device=dev).to(torch.bfloat16).contiguous(memory_format=cl)

    dy4, x4 = t(1024), t(1024)
    a2, dy1 = t(256), t(256)
    mean, inv = torch.zeros(1024, device=dev), torch.ones(1024, device=dev)
    w, b = torch.ones(1024, device=dev), torch.zeros(1024, device=dev)
    sdy, sdyx = torch.zeros(1024, device=dev), torch.zeros(1024, device=dev)
    dw3 = torch.zeros(1024, 256, 1, 1, device=dev).contiguous(memory_format=cl)
    w2t = (torch.randn(256, 256, 3, 3, device=dev) / 48).to(torch.bfloat16).contiguous(memory_format=cl)

    def A():
        _hip_bn.bwd_apply(dy4, x4, None, mean, inv, w, b, sdy, sdyx, float(n * hw * hw), 2, False)

    def Bk():
        _hip_conv.wgrad(dy4, a2, (1024, 256, 1, 1), (1, 1), (0, 0), (1, 1), out=dw3)

    def C():
        _hip_conv.igemm(dy1, w2t, (hw, hw), 1, 1, (1, 1), (1, 1))

    for f in (A, Bk, C):
        for _ in range(3):
            f()
    torch.cuda.synchronize()

    def alone(f, reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            f()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps

    ta, tb, tc = alone(A, args.iters), alone(Bk, args.iters), alone(C, args.iters)
    print(f"alone: bn_bwd_apply[M,1024] {ta * 1e3:.0f} us   wgrad 1x1 256->1024 {tb * 1e3:.0f} us   dgrad 3x3 256->256 {tc * 1e3:.0f} us")

    low = streams.aux_stream(dev)
    normal = torch.cuda.Stream(device=dev)

    def both(f, g, side, nf, ng):
        """nf launches of f on the current stream, ng of g on `side`, started together; wall time of the pair."""
        main = torch.cuda.current_stream(dev)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        side.wait_stream(main)
        with torch.cuda.stream(side):
            for _ in range(ng):
                g()
        for _ in range(nf):
            f()
        main.wait_stream(side)
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1)

    for name, f, g, tf, tg in (("bn_bwd_apply || wgrad", A, Bk, ta, tb), ("dgrad3x3 || wgrad", C, Bk, tc, tb),
                               ("dgrad3x3 || bn_bwd_apply", C, A, tc, ta)):
        # equal alone-time on both sides: n_f * t_f = n_g * t_g = ~args.iters * max
        tot = args.iters * max(tf, tg)
        nf, ng = max(1, round(tot / tf)), max(1, round(tot / tg))
        for sname, side in (("low-priority side stream", low), ("normal-priority side stream", normal)):
            both(f, g, side, 2, 2)
            wall = both(f, g, side, nf, ng)
            serial = nf * tf + ng * tg
            print(f"{name:28s} {sname:28s}: {nf} x {tf * 1e3:.0f} us + {ng} x {tg * 1e3:.0f} us = {serial:.2f} ms serial, "
                  f"{wall:.2f} ms together ({wall / serial:.2f} of serial; 0.50 = perfect overlap)")


if __name__ == "__main__":
    main()
