"""One training step of a model on the GPU kernels (variants) against fp32 on the CPU: gradient cosines.

    python tools/diag_model_step.py [model = deeplabv3plus_resnet18] [crop = 97]"""
import copy, sys, torch, torch.nn.functional as F
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
from test_gpu_hardening import cosine
from distributedpytorch_amd.models import build_model
from distributedpytorch_amd.ops import SoftmaxCrossEntropy, sepconv, _hip_conv
from distributedpytorch_amd.parallel import TrainEngine

name = sys.argv[1] if len(sys.argv) > 1 else "deeplabv3plus_resnet18"
size = int(sys.argv[2]) if len(sys.argv) > 2 else 97
kwargs = dict(output_stride=8) if "xception" in name else {}
torch.manual_seed(0)
m_cpu = build_model(name, 21, **kwargs).train()
with torch.no_grad():
    for m in m_cpu.modules():
        if hasattr(m, "running_mean"):
            m.weight.uniform_(0.7, 1.3); m.bias.normal_(0, 0.3)
damp = float(sys.argv[3]) if len(sys.argv) > 3 else 1.0
with torch.no_grad():
    for k, p in m_cpu.named_parameters():
        # the BatchNorm that closes a residual branch: small gamma = a network near the identity, as after
        # zero-init-residual training starts (random full-gain residual stacks are chaotic: see below)
        if k.endswith("bn3.weight") or (".rep." in k and k.endswith(".bn.weight") and not k.endswith("sep.bn.weight")):
            p.mul_(damp)
for m in m_cpu.modules():
    if isinstance(getattr(m, "p_drop", None), tuple): m.p_drop = (0.0, 0.0)
    elif hasattr(m, "p_drop"): m.p_drop = 0.0
g = torch.Generator().manual_seed(1)
x = torch.randn(4, 3, size, size, generator=g)
lab = torch.randint(0, 21, (4, 8, 8), generator=g)
lab = F.interpolate(lab[:, None].float(), size=(size, size), mode="nearest")[:, 0].long()
crit = SoftmaxCrossEntropy(255)
res = {}
for var, fold, prol in (("fused", True, True), ("plain", False, False)):
    sepconv._FOLD, _hip_conv._BN_PROLOGUE = fold, prol
    m_gpu = copy.deepcopy(m_cpu)
    eng = TrainEngine(m_gpu, crit, lr=0.0, momentum=0.0, weight_decay=0.0, device=torch.device("cuda", 0))
    out = eng.ddp(eng.prepare_input(x))
    l = crit(out, lab.cuda()); l.backward(); torch.cuda.synchronize()
    res[var] = (float(l.detach()), {s.name: s.param.grad.detach().float().cpu().clone() for s in eng.ddp.arena.slots})
m2 = copy.deepcopy(m_cpu)
out = m_cpu(x); l = crit(out, lab); l.backward()
ref = {k: p.grad.detach().float() for k, p in m_cpu.named_parameters()}
# a second fp32 CPU run whose INPUT is rounded to bf16: how far does a 2^-9 relative perturbation move the gradients?
out2 = m2(x.to(torch.bfloat16).float()); l2 = crit(out2, lab); l2.backward()
ref2 = {k: p.grad.detach().float() for k, p in m2.named_parameters()}
names = [k for k in ref if ref[k].dim() == 4 and float(ref[k].norm()) > 1e-7]
def summary(a, b):
    cs = [cosine(a[k], b[k]) for k in names]
    s = sorted(cs)
    return f"first {cs[0]:.4f} last {cs[-1]:.4f} median {s[len(s)//2]:.4f} min {s[0]:.4f}"
print("loss cpu", float(l.detach()), "fused", res["fused"][0], "plain", res["plain"][0], "cpu(bf16 input)", float(l2.detach()))
print("fused vs cpu     :", summary(res["fused"][1], ref))
print("plain vs cpu     :", summary(res["plain"][1], ref))
print("fused vs plain   :", summary(res["fused"][1], res["plain"][1]))
print("cpu bf16-in vs cpu:", summary(ref2, ref))
