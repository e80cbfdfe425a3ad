#!/usr/bin/env python
"""A/B of two conv_igemm2 tile configurations on one problem: outputs (and fused statistics / BN-backward sums) must
agree, times are printed side by side.   python tools/check_cfg.py --batch 64 --a 5 --b 21 728:728:65:1 256:256:33:3:2"""
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
    ap.add_argument("--a", type=int, default=5)
    ap.add_argument("--b", type=int, default=21)
    ap.add_argument("shapes", nargs="+", help="cin:cout:hw:k[:dil]")
    a = ap.parse_args()
    torch.manual_seed(0)
    bad = 0
    for sh in a.shapes:
        f = [int(v) for v in sh.split(":")]
        cin, cout, hw, k = f[:4]
        dil = f[4] if len(f) > 4 else 1
        pad = dil * (k - 1) // 2
        n = a.batch
        cl = torch.channels_last
        x = torch.randn(n, cin, hw, hw, device="cuda").to(torch.bfloat16).contiguous(memory_format=cl)
        w = (torch.randn(cout, cin, k, k, device="cuda") / (cin * k * k) ** 0.5).to(torch.bfloat16).contiguous(memory_format=cl)
        res = torch.randn(n, cout, hw, hw, device="cuda").to(torch.bfloat16).contiguous(memory_format=cl)
        ra = torch.randn(2, cout, device="cuda")
        reps = _hip_conv.stat_replicas(n * hw * hw)
        holder = {"x": res, "mean": torch.randn(cout, device="cuda") * 0.1, "invstd": torch.rand(cout, device="cuda") + 0.5,
                  "w": torch.rand(cout, device="cuda") + 0.5, "b": torch.randn(cout, device="cuda") * 0.1}
        res2 = torch.randn_like(res)

        def run(mode, cfg):
            if mode == "plain":
                return (_hip_conv.igemm(x, w, (hw, hw), 1, 1, (pad, pad), (dil, dil), cfg=cfg),)
            if mode == "stats":
                st = torch.zeros(reps, 2, cout, device="cuda")
                y = _hip_conv.igemm(x, w, (hw, hw), 1, 1, (pad, pad), (dil, dil), cfg=cfg, stats=st)
                return y, st.sum(0)
            if mode == "res":
                return (_hip_conv.igemm(x, w, (hw, hw), 1, 1, (pad, pad), (dil, dil), cfg=cfg, residual=res),)
            if mode == "resaff":
                return (_hip_conv.igemm(x, w, (hw, hw), 1, 1, (pad, pad), (dil, dil), cfg=cfg, residual=res,
                                        residual_affine=ra),)
            prev = _hip_conv._FORCED_CFG
            _hip_conv._FORCED_CFG = cfg
            try:
                y, sums = _hip_conv.dgrad_bn_fused(x, w, (hw, hw), 1, (pad, pad), (dil, dil), res2, holder)
                return y.clone(), sums.sum(0).clone()
            finally:
                _hip_conv._FORCED_CFG = prev

        for mode in ("plain", "stats", "res", "resaff", "bn"):
            oa, ob = run(mode, a.a), run(mode, a.b)
            torch.cuda.synchronize()
            errs = []
            for ta, tb in zip(oa, ob):
                ta, tb = ta.float(), tb.float()
                errs.append(float((ta - tb).abs().max() / ta.abs().max().clamp(min=1e-6)))
            ok = errs[0] == 0.0 and all(e < 2e-3 for e in errs[1:])    # same MFMA order: outputs bit-equal, atomics reorder sums
            bad += not ok
            ta_, tb_ = timeit(lambda: run(mode, a.a)), timeit(lambda: run(mode, a.b))
            print(f"{sh:20s} {mode:7s} cfg {a.a:2d}: {ta_:8.1f} us   cfg {a.b:2d}: {tb_:8.1f} us   ({ta_ / tb_:5.2f}x)  "
                  f"max rel diff {' '.join(f'{e:.2e}' for e in errs)}  {'ok' if ok else 'MISMATCH'}", flush=True)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
