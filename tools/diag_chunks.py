import sys, torch
sys.path.insert(0, '/root/repo')
from distributedpytorch_amd.models.xception import XceptionBlock
from distributedpytorch_amd.ops import _hip_conv
def cosine(a,b):
    a,b=a.double().flatten().cpu(),b.double().flatten().cpu(); return float((a@b)/(a.norm()*b.norm()).clamp_min(1e-30))
torch.manual_seed(4)
blk = torch.nn.Sequential(XceptionBlock(64, 64, 3, dilation=2), XceptionBlock(64, 128, 2, stride=1, dilation=1)).cuda().train()
x = torch.randn(6, 64, 33, 33, device='cuda').to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
gy=None; res={}
big=_hip_conv._OFFSET_LIMIT
for tag, limit in (("one", big), ("one2", big), ("chunks", 2*64*33*33*2+1)):
    _hip_conv._OFFSET_LIMIT = limit
    blk.zero_grad(set_to_none=True)
    xi = x.detach().clone(memory_format=torch.channels_last).requires_grad_(True)
    y = blk(xi)
    if gy is None: gy = torch.randn_like(y)
    y.backward(gy); torch.cuda.synchronize()
    res[tag] = (y.detach().float(), xi.grad.float(), {k: p.grad.detach().float().clone() for k,p in blk.named_parameters()})
for k in res["one"][2]:
    a=res["one"][2][k]; print(f"{k:40s} {tuple(a.shape)!s:18s} one-vs-one2 {cosine(a,res['one2'][2][k]):.5f}  one-vs-chunks {cosine(a,res['chunks'][2][k]):.5f}  norm {float(a.norm()):.3g}")
print("y", cosine(res['one'][0],res['chunks'][0]), "dx", cosine(res['one'][1],res['chunks'][1]), cosine(res['one'][1],res['one2'][1]))
