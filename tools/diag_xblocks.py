"""Tiny Xception stack: GPU variants against the CPU fp32 reference (prints cosines)."""
import copy, sys, torch
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import test_gpu_sepconv as T
from distributedpytorch_amd.ops import sepconv, _hip_conv

for stride in (1, 2):
    torch.manual_seed(5)
    net0 = T._TinyXception(stride).train()
    with torch.no_grad():
        for m in net0.modules():
            if hasattr(m, "running_mean"):
                m.weight.uniform_(0.6, 1.4); m.bias.normal_(0, 0.5)
    x = torch.randn(3, 16, 34, 30).to(torch.bfloat16)
    ref = copy.deepcopy(net0).float()
    with torch.no_grad():
        for p in ref.parameters():
            if p.dim() == 4:
                p.copy_(p.to(torch.bfloat16).float())
    xr = x.float().requires_grad_(True)
    yr = ref(xr)
    gy = torch.randn_like(yr).to(torch.bfloat16)
    yr.backward(gy.float())
    for name, fold, prol in (("all", True, True), ("no-prologue", True, False), ("no-fold", False, True),
                             ("plain", False, False)):
        sepconv._FOLD, _hip_conv._BN_PROLOGUE = fold, prol
        net = copy.deepcopy(net0).to("cuda")
        xh = x.to("cuda").contiguous(memory_format=torch.channels_last).requires_grad_(True)
        yh = net(xh)
        yh.backward(gy.to("cuda").contiguous(memory_format=torch.channels_last))
        torch.cuda.synchronize()
        worst, wk = 1.0, ""
        for (k, pg), (_, pr) in zip(net.named_parameters(), ref.named_parameters()):
            if k.endswith("sep.bn.bias") or float(pr.grad.norm()) < 1e-6 * float(pr.norm() + 1):
                continue
            c = T._cos(pg.grad, pr.grad)
            if c < worst:
                worst, wk = c, k
        print(f"stride {stride} {name:12s} y {T._cos(yh, yr):.6f} dx {T._cos(xh.grad, xr.grad):.6f} worst param {worst:.6f} ({wk})",
              flush=True)
