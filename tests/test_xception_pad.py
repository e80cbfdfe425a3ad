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


def test_padded_network_starts_from_the_same_weights_for_the_same_seed():
    """ADVICE (round 2): the 768-wide layers were initialised with their physical fan-out and another stretch of the
    generator.  Same seed -> same state_dict, and the generator ends where the unpadded constructor leaves it."""
    torch.manual_seed(123)
    a = AlignedXception(16, 3, pad_mid=False)
    after_a = torch.rand(4)
    torch.manual_seed(123)
    b = AlignedXception(16, 3, pad_mid=True)
    after_b = torch.rand(4)
    sa, sb = a.state_dict(), b.state_dict()
    for k in sa:
        assert torch.equal(sa[k], sb[k]), k
    assert torch.equal(after_a, after_b)
    for name, t, dims in b._padded_tensors():
        if t.dim() == 4:
            for d in dims:
                assert t.narrow(d, MID_CHANNELS, 40).abs().max().item() == 0.0, name


def test_optimizer_sidecar_moves_between_padded_and_unpadded_layouts(tmp_path):
    """ADVICE (round 2, medium): the momentum sidecar kept the padded arena layout, so a checkpoint of a native GPU
    run (768 wide) could not be resumed with --impl stock / on CPU (728 wide) and vice versa.  Format 2 stores the
    momentum per parameter NAME in the architecture's shapes."""
    import pytest
    from distributedpytorch_amd.models import build_model
    from distributedpytorch_amd.ops import SoftmaxCrossEntropy
    from distributedpytorch_amd.parallel import TrainEngine
    from distributedpytorch_amd.utils import load_checkpoint, save_checkpoint

    def engine(pad):
        torch.manual_seed(5)
        net = build_model("deeplabv3plus_xception", 21, output_stride=16)
        if pad != (net.backbone.mid_physical == 768):
            net.backbone = AlignedXception(16, 3, pad_mid=pad)
        return TrainEngine(net, SoftmaxCrossEntropy(255), lr=0.01, device=torch.device("cpu"))

    src = engine(True)
    assert src.model.backbone.mid_physical == 768
    x, y = torch.randn(2, 3, 65, 65), torch.randint(0, 21, (2, 65, 65))
    for _ in range(2):
        src.train_step(x, y)
    path = str(tmp_path / "x65.pth")
    save_checkpoint(path, src.ddp, src.optimizer, {"epoch": 0})
    side = torch.load(path + ".state", weights_only=True)["optimizer"]
    assert side["format"] == 2
    k = next(k for k in side["momentum_by_name"] if "middle.0" in k and k.endswith("pointwise.weight"))
    assert side["momentum_by_name"][k].shape[:2] == (MID_CHANNELS, MID_CHANNELS)
    from distributedpytorch_amd.ops import batchnorm as bn

    def step(eng):                      # the same dropout masks for every engine
        torch.manual_seed(77)
        bn._drop_calls = 0
        return float(eng.train_step(x, y))

    losses = []
    for pad in (False, True):
        dst = engine(pad)
        load_checkpoint(path, dst.ddp, dst.optimizer)
        ms, md = src.optimizer.state_dict()["momentum_by_name"], dst.optimizer.state_dict()["momentum_by_name"]
        assert list(ms) == list(md)
        for k in ms:
            assert torch.equal(ms[k], md[k]), (pad, k)
        losses.append(step(dst))        # and the next update is the same update
    losses.append(step(src))
    assert losses[0] == pytest.approx(losses[2], rel=1e-4) and losses[1] == pytest.approx(losses[2], rel=1e-5)
    for (k, a), (_, b) in zip(src.model.state_dict().items(), dst.model.state_dict().items()):
        if a.is_floating_point():
            assert torch.allclose(a, b, rtol=1e-4, atol=1e-6), k
    # a sidecar of another architecture is refused with a message, not a size mismatch inside copy_
    other = TrainEngine(build_model("fcn_resnet18", 21), SoftmaxCrossEntropy(255), lr=0.01, device=torch.device("cpu"))
    with pytest.raises(ValueError, match="does not match this model"):
        other.optimizer.load_state_dict(side)
