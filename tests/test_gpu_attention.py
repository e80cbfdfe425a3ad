"""DANet attention kernels (csrc/attention.hip) against the plain fp32 formulation (ops.attention.*_reference),
forward and backward, including ragged position counts and a spiked row that forces the online-softmax rescale.
Reference model: `/root/reference/train_pascal.py:86` (DANet), outputs `[1]`/`[2]` at `:258-260`."""
import os
import sys

import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pytestmark = pytest.mark.gpu


def _cl(t):
    return t.to(torch.bfloat16).cuda().contiguous(memory_format=torch.channels_last)


def _stats(got, ref):
    got, ref = got.float().flatten(), ref.float().flatten()
    cos = torch.nn.functional.cosine_similarity(got, ref, dim=0).item()
    rel = ((got - ref).norm() / ref.norm().clamp_min(1e-12)).item()
    return cos, rel


@pytest.mark.parametrize("hw,spike", [((64, 64), False), ((33, 33), False), ((17, 40), True)])
def test_position_attention_fwd_bwd(hw, spike):
    from distributedpytorch_amd.ops import attention as att
    h, w = hw
    n, c = 2, 512
    g = torch.Generator().manual_seed(h * 100 + w)
    q = torch.randn(n, 64, h, w, generator=g) * 0.4
    k = torch.randn(n, 64, h, w, generator=g) * 0.4
    if spike:
        # one key row aligned with one query row, late in the key sequence: the running maximum of that query jumps
        # by tens of units at a known block, the branch a bounded-random test never takes (rescale of O and l)
        q[0, :, 3, 5] = 2.0
        k[0, :, h - 1, w - 2] = 2.5
    v = torch.randn(n, c, h, w, generator=g)
    x = torch.randn(n, c, h, w, generator=g)
    gamma = torch.tensor([0.7])
    dy = torch.randn(n, c, h, w, generator=g)
    ins = [_cl(t).requires_grad_(True) for t in (q, k, v, x)]
    gam = gamma.cuda().requires_grad_(True)
    y = att.position_attention(*ins, gam)
    assert y.is_contiguous(memory_format=torch.channels_last) and y.dtype == torch.bfloat16
    y.backward(_cl(dy))
    # fp32 reference on the SAME bf16-rounded inputs
    rins = [t.detach().float().cpu().requires_grad_(True) for t in ins]
    rg = gamma.clone().requires_grad_(True)
    yr = att.position_attention_reference(*rins, rg)
    yr.backward(_cl(dy).float().cpu())
    cos, rel = _stats(y.cpu(), yr)
    assert cos > 0.9995 and rel < 0.02, ("y", cos, rel)
    for name, a, b in zip("qkvx", ins, rins):
        cos, rel = _stats(a.grad.cpu(), b.grad)
        assert cos > 0.999 and rel < 0.04, (name, cos, rel)
    assert abs(gam.grad.item() - rg.grad.item()) < 0.02 * abs(rg.grad.item()) + 0.5


@pytest.mark.parametrize("hw", [(64, 64), (33, 33), (9, 21)])
def test_channel_attention_fwd_bwd(hw):
    from distributedpytorch_amd.ops import attention as att
    h, w = hw
    n, c = 2, 512
    g = torch.Generator().manual_seed(h + 7 * w)
    # feature magnitudes such that the C x C energies spread over a few units (a softmax that is neither flat nor one-hot)
    x = torch.randn(n, c, h, w, generator=g) * (1.2 / (h * w) ** 0.25)
    gamma = torch.tensor([0.9])
    dy = torch.randn(n, c, h, w, generator=g)
    xi = _cl(x).requires_grad_(True)
    gam = gamma.cuda().requires_grad_(True)
    y = att.channel_attention(xi, gam)
    y.backward(_cl(dy))
    xr = xi.detach().float().cpu().requires_grad_(True)
    rg = gamma.clone().requires_grad_(True)
    yr = att.channel_attention_reference(xr, rg)
    yr.backward(_cl(dy).float().cpu())
    cos, rel = _stats(y.cpu(), yr)
    assert cos > 0.9995 and rel < 0.02, ("y", cos, rel)
    cos, rel = _stats(xi.grad.cpu(), xr.grad)
    assert cos > 0.999 and rel < 0.04, ("dx", cos, rel)
    assert abs(gam.grad.item() - rg.grad.item()) < 0.03 * abs(rg.grad.item()) + 0.5


def test_danet_step_matches_fp32_reference():
    """One DANet-R50 training step (4-channel input, 3 output heads; 512-channel head = the kernels' shapes) on the
    fused kernels vs the same model in fp32 on the CPU (plain PyTorch ops): loss and updated weights agree."""
    import copy

    from distributedpytorch_amd.models import build_model
    from distributedpytorch_amd.parallel import TrainEngine
    from distributedpytorch_amd.utils import SegmentationMultiLosses
    torch.manual_seed(0)
    m_cpu = build_model("danet_resnet50", 1, in_channels=4)
    with torch.no_grad():
        m_cpu.head.sa.gamma.fill_(0.5)
        m_cpu.head.sc.gamma.fill_(0.5)
    m_gpu = copy.deepcopy(m_cpu)
    g = torch.Generator().manual_seed(1)
    x = torch.randn(2, 4, 129, 129, generator=g)
    yt = (torch.rand(2, 1, 129, 129, generator=g) > 0.5).float()
    crit = SegmentationMultiLosses(1)
    m_cpu.eval()                      # no dropout / batch statistics noise in the comparison: BN uses its running stats
    m_gpu.eval()
    e_gpu = TrainEngine(m_gpu, crit, lr=0.0, momentum=0.0, weight_decay=0.0, device=torch.device("cuda", 0))
    e_gpu.model.eval()
    out_gpu = e_gpu.ddp(e_gpu.prepare_input(x))
    l_gpu = crit(out_gpu, yt.cuda())
    l_gpu.backward()
    out_cpu = m_cpu(x)
    l_cpu = crit(out_cpu, yt)
    l_cpu.backward()
    assert abs(float(l_gpu) - float(l_cpu)) < 0.03 * abs(float(l_cpu)) + 1e-3, (float(l_gpu), float(l_cpu))
    for o_g, o_c in zip(out_gpu, out_cpu):
        cos, rel = _stats(o_g.cpu(), o_c)
        assert cos > 0.995, ("head output", cos, rel)
    # gradients that reach the head through BOTH attention modules, against fp32.  (The query / key projections are
    # left to the unit tests above: in this random-init network the logits are in the thousands, the fp32 softmax
    # is exactly one-hot and its q/k gradients underflow to 1e-33 — nothing to compare against.)
    arena = e_gpu.ddp.arena
    for name in ("head.sa.gamma", "head.sc.gamma", "head.sa.value_conv.weight", "head.conv5c.0.weight",
                 "head.conv5a.0.weight"):
        p_cpu = dict(m_cpu.named_parameters())[name]
        slot = next(s for s in arena.slots if s.name == name)
        g_gpu = slot.param.grad.detach().float().cpu().reshape(-1)
        g_cpu = p_cpu.grad.detach().float().reshape(-1)
        if g_cpu.numel() == 1:
            assert abs(float(g_gpu) - float(g_cpu)) < 0.08 * abs(float(g_cpu)) + 1e-3, (name, float(g_gpu), float(g_cpu))
        else:
            cos, rel = _stats(g_gpu, g_cpu)
            assert cos > 0.98, (name, cos, rel)


@pytest.mark.parametrize("d,c", [(16, 128), (32, 256), (64, 768)])
def test_position_attention_other_widths_run_the_kernels(d, c):
    """any DANet that is not R50/R101-at-512 fell back to torch.bmm + softmax on the GPU without a word.
    Other widths (R18/R34 head: 16 q/k and 128 value channels) run the same kernels on zero-padded operands."""
    from distributedpytorch_amd.ops import attention as att
    n, h, w = 2, 24, 24
    g = torch.Generator().manual_seed(d + c)
    q, k = torch.randn(n, d, h, w, generator=g) * 0.5, torch.randn(n, d, h, w, generator=g) * 0.5
    v, x, dy = (torch.randn(n, c, h, w, generator=g) for _ in range(3))
    gamma = torch.tensor([0.6])
    ins = [_cl(t).requires_grad_(True) for t in (q, k, v, x)]
    gam = gamma.cuda().requires_grad_(True)
    calls = []
    orig = att._PositionAttention.apply
    att._PositionAttention.apply = staticmethod(lambda *a: (calls.append(tuple(t.shape[1] for t in a[:4])), orig(*a))[1])
    try:
        y = att.position_attention(*ins, gam)
    finally:
        att._PositionAttention.apply = orig
    assert calls and calls[0][0] == 64 and calls[0][2] % 256 == 0          # the kernels ran, on padded operands
    assert y.shape == (n, c, h, w)
    y.backward(_cl(dy))
    rins = [t.detach().float().cpu().requires_grad_(True) for t in ins]
    rg = gamma.clone().requires_grad_(True)
    yr = att.position_attention_reference(*rins, rg)
    yr.backward(_cl(dy).float().cpu())
    cos, rel = _stats(y.cpu(), yr)
    assert cos > 0.9995 and rel < 0.02, ("y", cos, rel)
    for name, a, b in zip("qkvx", ins, rins):
        assert a.grad.shape == b.grad.shape
        cos, rel = _stats(a.grad.cpu(), b.grad)
        assert cos > 0.999 and rel < 0.04, (name, cos, rel)


def test_unsupported_attention_shapes_raise_instead_of_falling_back():
    from distributedpytorch_amd.ops import attention as att
    gam = torch.zeros(1, device="cuda")
    with pytest.raises(NotImplementedError, match="multiple of 128"):
        att.channel_attention(_cl(torch.randn(1, 192, 8, 8)), gam)
    with pytest.raises(NotImplementedError, match="at most 64"):
        att.position_attention(_cl(torch.randn(1, 128, 8, 8)), _cl(torch.randn(1, 128, 8, 8)),
                               _cl(torch.randn(1, 256, 8, 8)), _cl(torch.randn(1, 256, 8, 8)), gam)
    # fp32 tensors / the stock-op backend still run the plain formulation (that is a choice, not a fallback)
    y = att.channel_attention(torch.randn(1, 192, 8, 8, device="cuda"), gam)
    assert y.shape == (1, 192, 8, 8)


def test_per_image_1x1_is_one_launch_and_matches_the_loop():
    """DANet leftovers: channel attention applied its per-image [C, C] matrices in a Python loop of N
    launches.  One launch with a per-image weight pointer; bit-identical to the loop (same kernel, same tile)."""
    from distributedpytorch_amd.ops import _hip_conv
    n, c, h, w = 5, 256, 16, 24                      # 384 positions per image: a multiple of 128, not of 256
    g = torch.Generator().manual_seed(3)
    x = _cl(torch.randn(n, c, h, w, generator=g))
    res = _cl(torch.randn(n, c, h, w, generator=g))
    wt = (torch.randn(n, c, c, generator=g) / 16).to(torch.bfloat16).cuda()
    got = _hip_conv.igemm_per_image(x, wt, res)
    ref = (torch.einsum("nchw,nkc->nkhw", x.float(), wt.float()) + res.float())
    cos, rel = _stats(got, ref)
    assert cos > 0.99999 and rel < 0.005, (cos, rel)
    loop = torch.empty_like(got)
    for b in range(n):
        _hip_conv.igemm(x[b:b + 1], wt[b].view(c, c, 1, 1), (h, w), 1, 1, (0, 0), (1, 1), cfg=1,
                        residual=res[b:b + 1], out=loop[b:b + 1])
    with _hip_conv.B.deterministic_mode(True):       # cfg 1 for the batched launch too
        one = _hip_conv.igemm_per_image(x, wt, res)
    assert torch.equal(one, loop)
    with pytest.raises(ValueError, match="multiple of the 128-row tile"):
        _hip_conv.igemm_per_image(_cl(torch.randn(2, 64, 9, 9)), wt[:2, :64, :64].contiguous())


def test_head_dropout_three_masks_one_pass():
    """models/danet.py: drop(sa), drop(sc), drop(sa + sc) — independent masks, keep rate 0.9, scale 1/0.9, and a
    backward that recomputes exactly the forward's masks."""
    from distributedpytorch_amd.ops import attention as att
    g = torch.Generator().manual_seed(5)
    sa = _cl(torch.randn(2, 512, 32, 32, generator=g).abs() + 0.5).requires_grad_(True)
    sc = _cl(torch.randn(2, 512, 32, 32, generator=g).abs() + 0.5).requires_grad_(True)
    torch.manual_seed(11)
    o = att.head_dropout(sa, sc, 0.1, True)
    srcs = (sa.detach().float(), sc.detach().float(), (sa.detach() + sc.detach()).float())
    keeps = []
    for out, src in zip(o, srcs):
        keep = out != 0
        keeps.append(keep)
        assert abs(keep.float().mean().item() - 0.9) < 0.005
        ratio = (out.float()[keep] / src[keep])
        assert (ratio - 1 / 0.9).abs().max().item() < 0.01
    for i in range(3):
        for j in range(i + 1, 3):
            assert 0.79 < (keeps[i] & keeps[j]).float().mean().item() < 0.83        # independent: 0.81
    gs = [_cl(torch.randn(2, 512, 32, 32, generator=g)) for _ in range(3)]
    torch.autograd.backward(o, gs)
    sc_ = 65536.0 / (65536 - round(0.1 * 65536))
    ra = (gs[0].float() * keeps[0] + gs[2].float() * keeps[2]) * sc_
    rc = (gs[1].float() * keeps[1] + gs[2].float() * keeps[2]) * sc_
    assert _stats(sa.grad, ra)[1] < 0.004 and _stats(sc.grad, rc)[1] < 0.004
    # another call draws other masks; eval mode is the identity
    o2 = att.head_dropout(sa, sc, 0.1, True)
    assert not torch.equal(o2[0] != 0, keeps[0])
    e = att.head_dropout(sa, sc, 0.1, False)
    assert e[0] is sa and e[1] is sc and torch.equal(e[2], sa + sc)
