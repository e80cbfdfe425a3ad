"""Print the error table of the folded / unfolded separable unit against the fp32 reference (GPU)."""
import sys, torch
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import test_gpu_sepconv as T

for case in T.CASES:
    n, cin, cout, h, w, stride, dil, with_res, relu = case
    torch.manual_seed(3)
    unit = T._Unit(cin, cout, stride, dil, relu_after=relu).to("cuda").train()
    with torch.no_grad():
        unit.sep.bn.weight.uniform_(0.5, 1.5); unit.sep.bn.bias.normal_(0, 0.5)
        unit.bn.weight.uniform_(0.5, 1.5); unit.bn.bias.normal_(0, 0.5)
        unit.sep.conv1.weight.mul_(torch.rand(cin, 1, 1, 1, device="cuda") * 3 + 0.2)
    x = (torch.randn(n, cin, h, w, device="cuda") + torch.randn(1, cin, 1, 1, device="cuda")).to(torch.bfloat16)
    x = x.contiguous(memory_format=torch.channels_last)
    ho = (h - 1) // stride + 1; wo = (w - 1) // stride + 1
    gy = torch.randn(n, cout, ho, wo, device="cuda").to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    res = torch.randn_like(gy) if with_res else None
    ref = T._reference(unit, x, gy, res, relu, dil, stride)
    for fold in (True, False):
        got = T._run_gpu(unit, x, gy, res, relu, fold)
        row = [f"y {T._rel(got[0], ref[0]):.4f}/{T._cos(got[0], ref[0]):.6f}",
               f"dx {T._rel(got[1], ref[1]):.4f}/{T._cos(got[1], ref[1]):.6f}"]
        for k in ("dw", "g1", "b1", "pw", "g2", "b2"):
            row.append(f"{k} {T._rel(got[3][k], ref[3][k]):.4f}/{T._cos(got[3][k], ref[3][k]):.6f}")
        for k in ("rm1", "rv1", "rm2", "rv2"):
            row.append(f"{k} {T._rel(got[4][k], ref[4][k]):.5f}")
        print(case, "fold" if fold else "plain", " ".join(row), flush=True)
