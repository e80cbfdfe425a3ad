"""Prediction without annotations on the MI355X: the kernels' path against the CPU forms of the same chain."""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

pytestmark = pytest.mark.gpu


def _cos(a, b):
    a, b = a.double().flatten(), b.double().flatten()
    return float((a @ b) / (a.norm() * b.norm() + 1e-30))


def _twin_models(name, nclass, in_channels=3):
    import copy
    from distributedpytorch_amd.models import build_model
    torch.manual_seed(0)
    m = build_model(name, nclass, in_channels=in_channels)
    with torch.no_grad():                       # running statistics a trained net would have: not all zeros / ones
        for mod in m.modules():
            if hasattr(mod, "running_var"):
                mod.running_var.uniform_(0.5, 1.5)
                mod.running_mean.uniform_(-0.2, 0.2)
    return m, copy.deepcopy(m)


def test_semseg_predictor_graph_equals_eager_and_tracks_fp32():
    from distributedpytorch_amd.inference import SemSegPredictor
    from distributedpytorch_amd.ops import _backend
    _backend.lib()                                           # (raises when the kernel library is missing)
    rng = np.random.default_rng(0)
    gpu_m, cpu_m = _twin_models("deeplabv3plus_resnet50", 21)
    imgs = [rng.integers(0, 256, s + (3,), dtype=np.uint8) for s in [(120, 200), (257, 257), (300, 90), (64, 64), (200, 333)]]
    # smooth images (a blur of the noise): bilinear resampling of white noise would compare rounding, not the chain
    imgs = [np.asarray(torch.nn.functional.avg_pool2d(torch.from_numpy(i).permute(2, 0, 1)[None].float(), 9, 1, 4)[0]
                       .permute(1, 2, 0)).astype(np.uint8) for i in imgs]
    graphed = SemSegPredictor(gpu_m, "cuda", crop=257, batch=2, graph=True)
    lg_g = [t.clone() for t in graphed.logits(imgs)]
    cls_g = graphed.predict(imgs)
    eager = SemSegPredictor(gpu_m, "cuda", crop=257, batch=2, graph=False)
    lg_e = eager.logits(imgs)
    for a, b, im, c in zip(lg_g, lg_e, imgs, cls_g):
        assert a.shape == (21,) + im.shape[:2] and c.shape == im.shape[:2]
        assert torch.equal(a, b)                             # the replay IS the eager forward
    assert len(graphed._fwd._graphs) == 1                    # the odd tail re-used the batch-2 graph
    ref = SemSegPredictor(cpu_m, "cpu", crop=257, batch=2).logits(imgs)
    for a, r in zip(lg_g, ref):
        assert _cos(a.cpu(), r) > 0.995


def test_instance_predictor_input_and_probabilities_track_the_cpu_chain():
    from distributedpytorch_amd.data import helpers
    from distributedpytorch_amd.inference import InstancePredictor
    rng = np.random.default_rng(1)
    gpu_m, cpu_m = _twin_models("danet_resnet50", 1, in_channels=4)
    shapes = [(180, 240), (256, 200), (150, 150)]
    imgs, pts = [], []
    for h, w in shapes:
        base = rng.integers(0, 256, (h // 8 + 1, w // 8 + 1, 3), dtype=np.uint8)
        imgs.append(np.kron(base, np.ones((8, 8, 1), dtype=np.uint8))[:h, :w])          # blocky: smooth inside blocks
        yy, xx = np.mgrid[:h, :w]
        mask = ((((yy - h * .5) / (h * .25)) ** 2 + ((xx - w * .45) / (w * .3)) ** 2) <= 1.0).astype(np.uint8)
        pts.append(helpers.extreme_points_fixed(mask))
    for zero_pad in (True, False):
        g = InstancePredictor(gpu_m, "cuda", crop=256, relax=50, zero_pad=zero_pad, batch=4)
        c = InstancePredictor(cpu_m, "cpu", crop=256, relax=50, zero_pad=zero_pad, batch=4)
        xg, cbg, _ = g.network_input(imgs, pts)
        xc, cbc, _ = c.network_input(imgs, pts)
        assert cbg.tolist() == cbc.tolist()
        assert xg.dtype == torch.bfloat16 and xg.is_contiguous(memory_format=torch.channels_last)
        d = (xg.float().cpu() - xc).abs()
        assert float(d[:, :3].mean()) < 0.6 and float(d[:, 3].mean()) < 1.0            # 0..255 scale, bf16 rounding
        pg, pc = g.predict(imgs, pts), c.predict(imgs, pts)
        for a, b, (h, w) in zip(pg, pc, shapes):
            assert a.shape == (h, w) and float(np.abs(a - b).mean()) < 0.02
            assert a.min() >= 0.0 and a.max() <= 1.0


def test_predict_cli_on_the_gpu(tmp_path, capsys):
    import json
    import predict
    rec = predict.main(["--model", "deeplabv3plus_resnet50", "--crop", "257", "--synthetic", "6", "--synthetic-size", "200x300",
                        "--batch", "4", "--output", str(tmp_path / "o")])
    assert rec["graph"] is True and rec["device"].startswith("cuda") and rec["steady"]
    assert len(os.listdir(tmp_path / "o")) == 6
    json.loads(capsys.readouterr().out.strip().splitlines()[-1])


# ---- the evaluation epilogue of the convolution kernel (csrc/conv_igemm2.hip, dp_conv_igemm2e) ----------------------
INFER_CASES = [
    # n, cin, cout, hw, k, stride, pad, dil, residual, relu, affine
    (2, 64, 256, 33, 1, 1, 0, 1, False, True, True),
    (2, 256, 64, 33, 1, 1, 0, 1, False, False, True),
    (3, 64, 64, 65, 3, 1, 1, 1, False, True, True),
    (2, 128, 128, 33, 3, 1, 2, 2, True, True, True),
    (2, 256, 512, 33, 1, 2, 0, 1, False, False, True),
    (2, 512, 2048, 17, 1, 1, 0, 1, True, True, True),
    (1, 256, 256, 33, 3, 1, 12, 12, False, True, True),
    (2, 64, 72, 20, 3, 2, 1, 1, True, False, True),
    (2, 64, 128, 16, 1, 1, 0, 1, True, True, False),
]


@pytest.mark.parametrize("case", INFER_CASES, ids=lambda c: "x".join(str(v) for v in c))
def test_inference_epilogue_matches_fp32(case):
    import torch.nn.functional as F
    from distributedpytorch_amd.ops import _hip_conv
    n, cin, cout, hw, k, stride, pad, dil, with_res, relu, with_aff = case
    g = torch.Generator().manual_seed(sum(case[:9]))
    x = torch.randn(n, cin, hw, hw, generator=g).to(torch.bfloat16)
    w = (torch.randn(cout, cin, k, k, generator=g) / (cin * k * k) ** 0.5).to(torch.bfloat16)
    aff = torch.stack([torch.rand(cout, generator=g) + 0.5, torch.randn(cout, generator=g)]) if with_aff else None
    ho = _hip_conv._out_size(hw, k, stride, pad, dil)
    res = torch.randn(n, cout, ho, ho, generator=g).to(torch.bfloat16) if with_res else None
    ref = F.conv2d(x.float(), w.float(), None, stride, pad, dil)
    if aff is not None:
        ref = ref * aff[0].view(1, -1, 1, 1) + aff[1].view(1, -1, 1, 1)
    if with_res:
        # the kernel rounds the convolution's result to bf16 (its LDS staging) before the identity is added
        ref = ref.to(torch.bfloat16).float() + res.float()
    if relu:
        ref = ref.clamp_min(0)
    dev = "cuda"
    y = _hip_conv.igemm_infer(x.to(dev).contiguous(memory_format=torch.channels_last),
                              w.to(dev).contiguous(memory_format=torch.channels_last), (ho, ho), stride, 1, (pad, pad),
                              (dil, dil), aff.to(dev).contiguous() if aff is not None else None, relu,
                              res.to(dev).contiguous(memory_format=torch.channels_last) if with_res else None)
    assert y is not None and y.shape == ref.shape and y.dtype == torch.bfloat16
    err = (y.float().cpu() - ref).abs()
    tol = 0.02 + 0.01 * ref.abs()
    assert bool((err <= tol).all()), f"max err {float(err.max()):.4f} at |ref| {float(ref.abs().max()):.2f}"
    if relu:
        assert float(y.min()) >= 0.0


@pytest.mark.parametrize("name", ["deeplabv3plus_resnet50", "deeplabv3plus_xception65"])
def test_eval_forward_fused_equals_two_pass_and_follows_the_weights(name):
    """The fused evaluation path against convolution + BatchNorm as two passes; and the cached (scale | shift) blocks
    follow a training step, a loaded state dict and an in-place edit — through an already captured graph too."""
    from distributedpytorch_amd.ops import SoftmaxCrossEntropy, conv as conv_mod, sepconv as sep_mod
    from distributedpytorch_amd.parallel import GraphedForward, TrainEngine
    torch.manual_seed(0)
    from distributedpytorch_amd.models import build_model
    model = build_model(name, 21)
    engine = TrainEngine(model, SoftmaxCrossEntropy(255), lr=0.01, momentum=0.9, weight_decay=1e-4, device=torch.device("cuda", 0))
    g = torch.Generator().manual_seed(1)
    xs = torch.randn(4, 3, 129, 129, generator=g)
    ys = torch.randint(0, 21, (4, 129, 129), generator=g)
    x_eval = engine.prepare_input(torch.randn(2, 3, 129, 129, generator=g))
    graphed = GraphedForward(engine.model)

    def two_pass():
        conv_mod.INFER_FUSE = sep_mod.EVAL_FOLD = False
        try:
            with torch.no_grad():
                for m in engine.model.modules():               # (fresh blocks: this is the reference)
                    if hasattr(m, "_affine_key"):
                        m._affine_key = None
                return engine.model(x_eval).float().clone()
        finally:
            conv_mod.INFER_FUSE = sep_mod.EVAL_FOLD = True

    def check(tag):
        engine.model.eval()
        with torch.no_grad():
            fused = engine.model(x_eval).float().clone()
        replay = graphed(x_eval).float().clone()
        ref = two_pass()
        assert _cos(fused, ref) > 0.9995, (tag, _cos(fused, ref))
        assert torch.equal(fused, replay), tag                  # the replay reads refreshed blocks
        return fused

    a = check("initial")
    engine.model.train()
    for _ in range(3):
        engine.train_step(xs, ys)                               # weights AND running statistics move (raw pointers)
    b = check("after training steps")
    assert _cos(a, b) < 0.9999                                   # (the network did change)
    sd = {k: v.clone() for k, v in engine.model.state_dict().items()}
    for k in sd:
        if k.endswith("running_var"):
            sd[k] = sd[k] * 1.7
    engine.model.load_state_dict(sd)
    c = check("after load_state_dict")
    assert _cos(b, c) < 0.9999
    with torch.no_grad():
        for m in engine.model.modules():
            if hasattr(m, "running_mean") and m.running_mean is not None:
                m.running_mean.add_(0.05)                       # an in-place edit while in evaluation mode
    d = check("after an in-place edit")
    assert _cos(c, d) < 0.9999


def test_predict_cli_two_ranks_sharing_the_gpu(tmp_path):
    """`predict.py --gpus 2` as the script starts it (two processes on the box's one GPU, gloo rendezvous): the images are
    dealt between the ranks, every map is written once, rank 0 prints the job's line."""
    import json
    import subprocess
    env = dict(os.environ, DPAMD_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = tmp_path / "o"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "predict.py"), "--model", "deeplabv3plus_resnet50", "--crop", "257",
                        "--synthetic", "10", "--synthetic-size", "120x200", "--batch", "4", "--gpus", "2", "--output", str(out)],
                       capture_output=True, text=True, timeout=240, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    rec = json.loads(lines[0])
    assert rec["images"] == 10 and rec["n_gpus"] == 2 and rec["graph"] is True
    assert len(os.listdir(out)) == 10
