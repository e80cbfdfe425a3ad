"""The 728-channel flow of Aligned Xception-65 stored 768 wide (`models/xception.py` docstring): the padded
network IS the 728-channel network — same outputs, same updates, padding weights stay exact zeros — and its
``state_dict`` keeps the architecture's shapes in both directions."""
import torch

from distributedpytorch_amd.models.xception import AlignedXception, MID_CHANNELS


def _pair():
    torch.manual_seed(0)
    a = AlignedXception(8, 3, pad_mid=False)
    b = AlignedXception(8, 3, pad_mid=True)
    assert b.mid_physical == 768 and a.mid_physical == MID_CHANNELS
    assert b.load_state_dict(a.state_dict()).missing_keys == []
    return a, b


def test_state_dict_speaks_logical_shapes():
    a, b = _pair()
    sa, sb = a.state_dict(), b.state_dict()
    assert list(sa.keys()) == list(sb.keys())
    for k in sa:
        assert sa[k].shape == sb[k].shape, k
        assert torch.equal(sa[k], sb[k]), k
    assert b.logical_num_parameters() == sum(p.numel() for p in a.parameters())
    assert sum(p.numel() for p in b.parameters()) > b.logical_num_parameters()


def test_padded_flow_trains_like_the_728_channel_network():
    a, b = _pair()
    a.train(), b.train()
    opts = [torch.optim.SGD(m.parameters(), lr=0.05, momentum=0.9, weight_decay=1e-3) for m in (a, b)]
    torch.manual_seed(1)
    for step in range(3):
        x = torch.randn(2, 3, 65, 65)
        outs = []
        for m, opt in zip((a, b), opts):
            opt.zero_grad()
            y, low = m(x)
            (y.square().mean() + low.square().mean()).backward()
            opt.step()
            outs.append(y.detach())
        assert torch.allclose(outs[0], outs[1], rtol=1e-4, atol=1e-5), step
    # padding weights: exact zeros after momentum + weight-decay steps (every gradient into them is a product with 0)
    for name, t, dims in b._padded_tensors():
        if t.dim() == 4:
            for d in dims:
                assert t.narrow(d, MID_CHANNELS, 40).abs().max().item() == 0.0, name
    sa, sb = a.state_dict(), b.state_dict()
    for k in sa:
        if sa[k].is_floating_point():
            assert torch.allclose(sa[k], sb[k], rtol=1e-3, atol=1e-5), k
