"""Where does the run-to-run spread of the gradients come from?  (VERDICT r2, weak #1.)

For the two stacks of tests/test_gpu_hardening.py (a two-block Xception stack on the rolling-window depthwise
kernels with the folded separable units; a dense conv-BN-ReLU stack) this prints, per parameter:

* the gradient of an fp64 CPU run of the SAME module code (torch path in double: the yardstick),
* the relative error ``|g - g64| / |g64|`` of the kernels' gradient — default mode and deterministic mode,
  folded and unfolded separable path, one launch and image chunks,
* the cosine between two identical launches and between one launch and image chunks, in both modes, and whether
  two identical launches are BIT-identical.

A parameter whose fp64 gradient is tiny next to its neighbours' is ill-conditioned: a BatchNorm scale whose
output only meets ReLU -> depthwise convolution -> BatchNorm is (nearly) scale-invariant when its shift is 0 —
relu(g x) = g relu(x), the depthwise convolution is per-channel linear and the next BatchNorm divides the scale
out — so its true gradient is ~0 (eps and the zero padding at the border leave a remainder) and what the bf16
kernels compute for it is rounding noise.  ``--bias`` re-draws the BatchNorm shifts / scales (N(0, 0.5) /
U(0.6, 1.4)): the invariance is gone and every gradient is well-conditioned.

    python tools/diag_nondeterminism.py [--bias] [--dense] > profiles/r3_nondeterminism_*.md
"""
import argparse
import copy
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from distributedpytorch_amd import ops  # noqa: E402
from distributedpytorch_amd.ops import _backend as B  # noqa: E402
from distributedpytorch_amd.ops import _hip_conv, sepconv  # noqa: E402


def cosine(a, b):
    a, b = a.double().flatten().cpu(), b.double().flatten().cpu()
    return float((a @ b) / (a.norm() * b.norm()).clamp_min(1e-300))


def rel(a, ref):
    a, ref = a.double().cpu(), ref.double().cpu()
    return float((a - ref).norm() / ref.norm().clamp_min(1e-300))


def build(dense: bool, bias: bool):
    if dense:
        torch.manual_seed(3)

        class Net(torch.nn.Module):
            def __init__(self):
                super().__init__()
                self.c1 = ops.Conv2d(32, 64, 3, padding=1, bias=False); self.b1 = ops.BatchNorm2d(64)
                self.c2 = ops.Conv2d(64, 64, 1, bias=False); self.b2 = ops.BatchNorm2d(64)
                self.c3 = ops.Conv2d(64, 32, 3, padding=2, dilation=2, bias=False); self.b3 = ops.BatchNorm2d(32)

            def forward(self, x):
                a = ops.conv_bn_act(self.c1, self.b1, x, relu=True)
                b = ops.conv_bn_act(self.c2, self.b2, a, relu=True, residual=a)
                return ops.conv_bn_act(self.c3, self.b3, b, relu=True)

        net, cin = Net(), 32
    else:
        from distributedpytorch_amd.models.xception import XceptionBlock
        torch.manual_seed(4)
        net = torch.nn.Sequential(XceptionBlock(64, 64, 3, dilation=2), XceptionBlock(64, 128, 2, stride=1, dilation=1))
        cin = 64
    if bias:
        with torch.no_grad():
            for m in net.modules():
                if hasattr(m, "running_mean"):
                    m.weight.uniform_(0.6, 1.4); m.bias.normal_(0, 0.5)
    x = torch.randn(6, cin, 33, 33).to(torch.bfloat16)
    return net.train(), x, cin


def run_gpu(net, x, gy, chunks: bool, cin: int):
    big = _hip_conv._OFFSET_LIMIT
    _hip_conv._OFFSET_LIMIT = (2 * cin * 33 * 33 * 2 + 1) if chunks else big
    try:
        net.zero_grad(set_to_none=True)
        xi = x.detach().clone(memory_format=torch.channels_last).requires_grad_(True)
        y = net(xi)
        if gy is None:
            gy = torch.randn_like(y).contiguous(memory_format=torch.channels_last)
        y.backward(gy)
        torch.cuda.synchronize()
        return gy, {k: p.grad.detach().float().clone() for k, p in net.named_parameters()}
    finally:
        _hip_conv._OFFSET_LIMIT = big


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--bias", action="store_true", help="random BatchNorm shifts / scales (breaks the scale invariance)")
    ap.add_argument("--dense", action="store_true", help="the dense conv-BN-ReLU stack instead of the Xception blocks")
    args = ap.parse_args()
    net, x, cin = build(args.dense, args.bias)
    gnet = copy.deepcopy(net).cuda()
    xg = x.cuda().contiguous(memory_format=torch.channels_last)
    gy, _ = run_gpu(gnet, xg, None, False, cin)             # also warms the tile choices up
    # fp64 yardstick: the same modules on the torch path in double, convolution weights rounded to bf16 as the
    # kernels see them, the same bf16 input and output gradient
    ref = copy.deepcopy(net).double()
    with torch.no_grad():
        for p in ref.parameters():
            if p.dim() == 4:
                p.copy_(p.to(torch.bfloat16).double())
    x64 = x.double().requires_grad_(True)
    y64 = ref(x64)
    y64.backward(gy.double().cpu())
    g64 = {k: p.grad.detach() for k, p in ref.named_parameters()}

    results = {}
    variants = [("fold", True)] if args.dense else [("fold", True), ("plain", False)]
    for vname, fold in variants:
        old = sepconv._FOLD
        sepconv._FOLD = fold
        try:
            for mode, det in (("default", False), ("determ", True)):
                with B.deterministic_mode(det):
                    for tag, ch in (("one", False), ("again", False), ("chunks", True)):
                        _, g = run_gpu(gnet, xg, gy, ch, cin)
                        results[(vname, mode, tag)] = g
        finally:
            sepconv._FOLD = old

    title = ("dense conv-BN-ReLU stack" if args.dense else "two Xception blocks (folded separable units)") + \
            (", random BatchNorm shifts" if args.bias else ", default BatchNorm init (shift 0)")
    print(f"# Gradient reproducibility: {title}\n")
    print("`|g64|` = norm of the fp64 gradient; `err` = |g - g64| / |g64|; `cos(a,b)` between two runs of the kernels; "
          "`bit` = two identical launches bit-identical.\n")
    for vname, _ in variants:
        print(f"## separable path: {vname}\n" if not args.dense else "## dense\n")
        print("| parameter | shape | \\|g64\\| | err default | err determ. | default cos(one,again) | default cos(one,chunks) | "
              "determ. bit | determ. cos(one,chunks) |")
        print("|---|---|---:|---:|---:|---:|---:|:---:|---:|")
        worst_bit = True
        for k in g64:
            d1, d2, dc = (results[(vname, "default", t)][k] for t in ("one", "again", "chunks"))
            e1, e2, ec = (results[(vname, "determ", t)][k] for t in ("one", "again", "chunks"))
            bit = bool(torch.equal(e1, e2))
            worst_bit &= bit
            print(f"| {k} | {tuple(d1.shape)} | {float(g64[k].norm()):.4g} | {rel(d1, g64[k]):.3g} | {rel(e1, g64[k]):.3g} | "
                  f"{cosine(d1, d2):.6f} | {cosine(d1, dc):.6f} | {'yes' if bit else 'NO'} | {cosine(e1, ec):.6f} |")
        print(f"\nall parameters bit-identical between two deterministic launches: **{worst_bit}**\n")


if __name__ == "__main__":
    main()
