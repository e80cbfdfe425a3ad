#!/usr/bin/env python
"""Per-layer convolution micro-benchmark: hand-written implicit-GEMM kernels vs MIOpen (stock
``F.conv2d``), for every distinct conv shape of DeepLabv3+ R101 @ 513² (OS 16).

    python tools/bench_conv.py [--batch 16] [--stock] [--only fwd|dgrad|wgrad]

Prints µs and achieved TFLOP/s per pass; totals are weighted by how often each shape occurs in the
network.  All passes of one shape are timed interleaved in ONE process (HIP events, median of N).
"""
import argparse
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from distributedpytorch_amd.ops import _hip_conv  # noqa: E402

# (name, count, cin, cout, hw_in, k, stride, pad, dil)
R101_DEEPLAB = [
    ("stem7x7s2", 1, 3, 64, 513, 7, 2, 3, 1),
    ("l1.conv1.first", 1, 64, 64, 129, 1, 1, 0, 1),
    ("l1.conv2", 3, 64, 64, 129, 3, 1, 1, 1),
    ("l1.conv3", 3, 64, 256, 129, 1, 1, 0, 1),
    ("l1.down", 1, 64, 256, 129, 1, 1, 0, 1),
    ("l1.conv1", 2, 256, 64, 129, 1, 1, 0, 1),
    ("l2.conv1.first", 1, 256, 128, 129, 1, 1, 0, 1),
    ("l2.conv2.s2", 1, 128, 128, 129, 3, 2, 1, 1),
    ("l2.down.s2", 1, 256, 512, 129, 1, 2, 0, 1),
    ("l2.conv3", 4, 128, 512, 65, 1, 1, 0, 1),
    ("l2.conv1", 3, 512, 128, 65, 1, 1, 0, 1),
    ("l2.conv2", 3, 128, 128, 65, 3, 1, 1, 1),
    ("l3.conv1.first", 1, 512, 256, 65, 1, 1, 0, 1),
    ("l3.conv2.s2", 1, 256, 256, 65, 3, 2, 1, 1),
    ("l3.down.s2", 1, 512, 1024, 65, 1, 2, 0, 1),
    ("l3.conv3", 23, 256, 1024, 33, 1, 1, 0, 1),
    ("l3.conv1", 22, 1024, 256, 33, 1, 1, 0, 1),
    ("l3.conv2", 22, 256, 256, 33, 3, 1, 1, 1),
    ("l4.conv1.first", 1, 1024, 512, 33, 1, 1, 0, 1),
    ("l4.down", 1, 1024, 2048, 33, 1, 1, 0, 1),
    ("l4.conv2.d2", 1, 512, 512, 33, 3, 1, 2, 2),
    ("l4.conv2.d4", 1, 512, 512, 33, 3, 1, 4, 4),
    ("l4.conv2.d8", 1, 512, 512, 33, 3, 1, 8, 8),
    ("l4.conv3", 3, 512, 2048, 33, 1, 1, 0, 1),
    ("l4.conv1", 2, 2048, 512, 33, 1, 1, 0, 1),
    ("aspp1", 1, 2048, 256, 33, 1, 1, 0, 1),
    ("aspp.d6", 1, 2048, 256, 33, 3, 1, 6, 6),
    ("aspp.d12", 1, 2048, 256, 33, 3, 1, 12, 12),
    ("aspp.d18", 1, 2048, 256, 33, 3, 1, 18, 18),
    ("aspp.proj", 1, 1280, 256, 33, 1, 1, 0, 1),
    ("dec.conv1", 1, 256, 48, 129, 1, 1, 0, 1),
    ("dec.last0", 1, 304, 256, 129, 3, 1, 1, 1),
    ("dec.last4", 1, 256, 256, 129, 3, 1, 1, 1),
    ("dec.cls", 1, 256, 21, 129, 1, 1, 0, 1),
]


# Xception-65 OS8 (728-wide flows stored 768 wide): the pointwise halves of the separable units
X65_OS8 = [
    ("entry.pw128", 2, 128, 128, 257, 1, 1, 0, 1),
    ("entry.pw256", 2, 256, 256, 129, 1, 1, 0, 1),
    ("mid.pw768", 52, 768, 768, 65, 1, 1, 0, 1),
    ("exit.pw1024", 1, 768, 1024, 65, 1, 1, 0, 1),
    ("exit.pw1536", 2, 1536, 1536, 65, 1, 1, 0, 1),
    ("exit.pw2048", 1, 1536, 2048, 65, 1, 1, 0, 1),
]
LAYERS = {"r101": R101_DEEPLAB, "x65": X65_OS8}


def timeit(fn, iters=5, warm=2, inner=10):
    """Median µs per call; ``inner`` calls are queued back to back between the two events so the
    host-side launch cost of one call hides behind the previous kernel."""
    for _ in range(warm):
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
    ts.sort()
    return ts[len(ts) // 2]


def sweep(args):
    """fwd (with fused statistics) and dgrad time of every requested tile config, per layer."""
    cfgs = [int(c) for c in args.cfgs.split(",")]
    dev = "cuda"
    print(f"{'layer':16s} {'cnt':>3s} {'M':>7s} {'N':>5s} {'K':>6s} | " + " ".join(f"f{c:<6d}" for c in cfgs) + " | "
          + " ".join(f"d{c:<6d}" for c in cfgs) + " | best fwd/dgrad")
    tot_f = {c: 0.0 for c in cfgs}
    tot_d = {c: 0.0 for c in cfgs}
    best_f = best_d = 0.0
    for name, cnt, cin, cout, hw, k, stride, pad, dil in LAYERS[args.net]:
        if (args.filter and args.filter not in name) or k == 7:
            continue
        n = args.batch
        ho = (hw + 2 * pad - dil * (k - 1) - 1) // stride + 1
        cin_p, cout_p = (cin + 7) // 8 * 8, (cout + 7) // 8 * 8
        x = torch.randn(n, cin_p, hw, hw, device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
        w = (torch.randn(cout_p, cin_p, k, k, device=dev) / (cin * k * k) ** 0.5).to(torch.bfloat16) \
            .contiguous(memory_format=torch.channels_last)
        dy = torch.randn(n, cout_p, ho, ho, device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
        wt = _hip_conv.weight_transpose(w)
        pad_t = dil * (k - 1) - pad
        reps = _hip_conv.stat_replicas(n * ho * ho)
        stats = torch.zeros(reps, 2, cout_p, device=dev)
        tf, td = {}, {}
        for c in cfgs:
            tf[c] = timeit(lambda: _hip_conv.igemm(x, w, (ho, ho), stride, 1, (pad, pad), (dil, dil), None, stats, cfg=c))
            td[c] = timeit(lambda: _hip_conv.igemm(dy, wt, (hw, hw), 1, stride, (pad_t, pad_t), (dil, dil), cfg=c))
            tot_f[c] += cnt * tf[c]
            tot_d[c] += cnt * td[c]
        bf, bd = min(tf, key=tf.get), min(td, key=td.get)
        best_f += cnt * tf[bf]
        best_d += cnt * td[bd]
        print(f"{name:16s} {cnt:3d} {n * ho * ho:7d} {cout_p:5d} {cin_p * k * k:6d} | "
              + " ".join(f"{tf[c]:7.1f}" for c in cfgs) + " | " + " ".join(f"{td[c]:7.1f}" for c in cfgs)
              + f" | {bf} {bd}", flush=True)
    print("TOTAL ms       " + " " * 27 + " ".join(f"{tot_f[c] / 1e3:7.2f}" for c in cfgs) + " | "
          + " ".join(f"{tot_d[c] / 1e3:7.2f}" for c in cfgs) + f" | best-per-layer fwd {best_f / 1e3:.2f} dgrad {best_d / 1e3:.2f}")


def sweep_wgrad(args):
    splits = [int(c) for c in args.splits.split(",")]
    dev = "cuda"
    print(f"{'layer':16s} {'cnt':>3s} {'P':>7s} {'Cout':>5s} {'RSC':>6s} | " + " ".join(f"s{c:<6d}" for c in splits))
    tot = {c: 0.0 for c in splits}
    for name, cnt, cin, cout, hw, k, stride, pad, dil in LAYERS[args.net]:
        if args.filter and args.filter not in name:
            continue
        n = args.batch
        ho = (hw + 2 * pad - dil * (k - 1) - 1) // stride + 1
        cin_p, cout_p = (cin + 7) // 8 * 8, (cout + 7) // 8 * 8
        x = torch.randn(n, cin_p, hw, hw, device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
        dy = torch.randn(n, cout_p, ho, ho, device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
        out = torch.zeros(cout_p, cin_p, k, k, device=dev).contiguous(memory_format=torch.channels_last)
        ts = {}
        for c in splits:
            ts[c] = timeit(lambda: _hip_conv.wgrad(dy, x, (cout_p, cin_p, k, k), (stride, stride), (pad, pad),
                                                   (dil, dil), out=out, splits=c))
            tot[c] += cnt * ts[c]
        print(f"{name:16s} {cnt:3d} {n * ho * ho:7d} {cout_p:5d} {cin_p * k * k:6d} | "
              + " ".join(f"{ts[c]:7.1f}" for c in splits), flush=True)
    print("TOTAL ms" + " " * 35 + " ".join(f"{tot[c] / 1e3:7.2f}" for c in splits))


def lib_gemm(args):
    """1x1 stride-1 layers: [M, K] x [K, N] in bf16.  Forward with the statistics epilogue the step uses, the data
    gradient plain, the weight gradient into an fp32 [N, K] — against torch.matmul on the same 2-D views."""
    dev = "cuda"
    print(f"{'layer':16s} {'cnt':>3s} {'M':>8s} {'N':>5s} {'K':>5s} | {'fwd':>7s} {'lib':>7s} {'TF':>5s} {'libTF':>5s} | "
          f"{'dgrad':>7s} {'lib':>7s} {'TF':>5s} {'libTF':>5s} | {'wgrad':>7s} {'lib':>7s} {'TF':>5s} {'libTF':>5s}")
    tot = [0.0] * 6
    for name, cnt, cin, cout, hw, k, stride, pad, dil in LAYERS[args.net]:
        if (args.filter and args.filter not in name) or k != 1 or stride != 1 or cin % 8 or cout % 8:
            continue
        n = args.batch
        m = n * hw * hw
        x = torch.randn(n, cin, hw, hw, device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
        w = (torch.randn(cout, cin, 1, 1, device=dev) / cin ** 0.5).to(torch.bfloat16) \
            .contiguous(memory_format=torch.channels_last)
        dy = torch.randn(n, cout, hw, hw, device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
        wt = _hip_conv.weight_transpose(w)
        stats = torch.zeros(_hip_conv.stat_replicas(m), 2, cout, device=dev)
        out = torch.zeros(cout, cin, 1, 1, device=dev).contiguous(memory_format=torch.channels_last)
        x2, dy2, w2 = x.permute(0, 2, 3, 1).reshape(m, cin), dy.permute(0, 2, 3, 1).reshape(m, cout), w.view(cout, cin)
        w2t, dy2t = w2.t().contiguous(), dy2.t()
        gf = 2.0 * m * cin * cout / 1e9
        t = [timeit(lambda: _hip_conv.igemm(x, w, (hw, hw), 1, 1, (0, 0), (1, 1), None, stats)),
             timeit(lambda: torch.matmul(x2, w2t)),
             timeit(lambda: _hip_conv.igemm(dy, wt, (hw, hw), 1, 1, (0, 0), (1, 1))),
             timeit(lambda: torch.matmul(dy2, w2)),
             timeit(lambda: _hip_conv.wgrad(dy, x, (cout, cin, 1, 1), (1, 1), (0, 0), (1, 1), out=out)),
             timeit(lambda: torch.matmul(dy2t, x2))]
        t_plain = timeit(lambda: _hip_conv.igemm(x, w, (hw, hw), 1, 1, (0, 0), (1, 1)))
        for i in range(6):
            tot[i] += cnt * t[i]
        print(f"{name:16s} {cnt:3d} {m:8d} {cout:5d} {cin:5d} | " + " | ".join(
            f"{t[i]:7.1f} {t[i + 1]:7.1f} {gf / t[i] * 1e3:5.0f} {gf / t[i + 1] * 1e3:5.0f}" for i in (0, 2, 4))
              + f" | fwd without the statistics epilogue {t_plain:7.1f}", flush=True)
        del x, dy, x2, dy2
    print("TOTAL weighted ms: " + "  ".join(f"{nm} {tot[i] / 1e3:.2f} lib {tot[i + 1] / 1e3:.2f}"
                                             for nm, i in (("fwd", 0), ("dgrad", 2), ("wgrad", 4))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--stock", action="store_true", help="also time MIOpen through F.conv2d")
    ap.add_argument("--filter", default="")
    ap.add_argument("--net", default="r101", choices=sorted(LAYERS))
    ap.add_argument("--splits", default="", help="comma list of wgrad split-K counts to sweep (0 = auto)")
    ap.add_argument("--cfgs", default="", help="comma list of conv_igemm2 tile configs to sweep (0 = gen-1 kernel)")
    ap.add_argument("--lib-gemm", action="store_true",
                    help="yardstick: the 1x1 stride-1 layers as plain GEMMs through torch.matmul (hipBLASLt), same "
                         "operands, next to the hand-written kernels (which also carry the fused epilogues)")
    args = ap.parse_args()
    if args.cfgs:
        sweep(args)
        return
    if args.splits:
        sweep_wgrad(args)
        return
    if args.lib_gemm:
        lib_gemm(args)
        return
    dev = "cuda"
    tot = {"fwd": 0.0, "dgrad": 0.0, "wgrad": 0.0, "sfwd": 0.0, "sbwd": 0.0}
    tot_fl = 0.0
    print(f"{'layer':16s} {'cnt':>3s} {'GF':>7s} | {'fwd us':>8s} {'TF':>6s} | {'dgrad us':>8s} {'TF':>6s} | "
          f"{'wgrad us':>8s} {'TF':>6s}" + (" | stock fwd  bwd(us)" if args.stock else ""))
    for name, cnt, cin, cout, hw, k, stride, pad, dil in LAYERS[args.net]:
        if args.filter and args.filter not in name:
            continue
        n = args.batch
        ho = (hw + 2 * pad - dil * (k - 1) - 1) // stride + 1
        cin_p, cout_p = (cin + 7) // 8 * 8, (cout + 7) // 8 * 8
        x = torch.randn(n, cin_p, hw, hw, device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
        w = (torch.randn(cout_p, cin_p, k, k, device=dev) / (cin * k * k) ** 0.5).to(torch.bfloat16) \
            .contiguous(memory_format=torch.channels_last)
        dy = torch.randn(n, cout_p, ho, ho, device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
        wt = _hip_conv.weight_transpose(w)
        pad_t = dil * (k - 1) - pad
        gf = 2.0 * n * ho * ho * cout * cin * k * k / 1e9
        t_f = timeit(lambda: _hip_conv.igemm(x, w, (ho, ho), stride, 1, (pad, pad), (dil, dil)))
        t_d = timeit(lambda: _hip_conv.igemm(dy, wt, (hw, hw), 1, stride, (pad_t, pad_t), (dil, dil)))
        t_w = timeit(lambda: _hip_conv.wgrad(dy, x, (cout_p, cin_p, k, k), (stride, stride), (pad, pad), (dil, dil)))
        line = (f"{name:16s} {cnt:3d} {gf:7.1f} | {t_f:8.1f} {gf / t_f * 1e3:6.1f} | {t_d:8.1f} {gf / t_d * 1e3:6.1f} | "
                f"{t_w:8.1f} {gf / t_w * 1e3:6.1f}")
        tot['fwd'] += cnt * t_f
        tot['dgrad'] += cnt * t_d
        tot['wgrad'] += cnt * t_w
        tot_fl += cnt * gf
        if args.stock:
            xs = x.detach().clone().requires_grad_(True)
            ws = w.detach().clone().requires_grad_(True)
            t_sf = timeit(lambda: F.conv2d(xs, ws, None, stride, pad, dil))

            def fb():
                y = F.conv2d(xs, ws, None, stride, pad, dil)
                y.backward(dy)
            t_sb = timeit(fb) - t_sf
            tot["sfwd"] += cnt * t_sf
            tot["sbwd"] += cnt * t_sb
            line += f" | {t_sf:8.1f} {t_sb:8.1f}"
        print(line, flush=True)
    print(f"TOTAL weighted: fwd {tot['fwd'] / 1e3:.2f} ms ({tot_fl / tot['fwd'] * 1e3:.0f} TF)  "
          f"dgrad {tot['dgrad'] / 1e3:.2f} ms ({tot_fl / tot['dgrad'] * 1e3:.0f} TF)  "
          f"wgrad {tot['wgrad'] / 1e3:.2f} ms ({tot_fl / tot['wgrad'] * 1e3:.0f} TF)  [{tot_fl / 1e3:.2f} TFLOP per pass]")
    if args.stock:
        print(f"STOCK (MIOpen): fwd {tot['sfwd'] / 1e3:.2f} ms  bwd(dgrad+wgrad) {tot['sbwd'] / 1e3:.2f} ms")


if __name__ == "__main__":
    main()
