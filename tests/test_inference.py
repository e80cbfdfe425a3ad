"""Prediction without annotations (distributedpytorch_amd.inference, predict.py) on the CPU forms of the ops."""
import json
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from distributedpytorch_amd.data import helpers  # noqa: E402
from distributedpytorch_amd.inference import InstancePredictor, SemSegPredictor, voc_palette  # noqa: E402
from distributedpytorch_amd.models import build_model  # noqa: E402
from distributedpytorch_amd.ops.augment import IMAGENET_MEAN, IMAGENET_STD, instance_augment  # noqa: E402


def _images(rng, shapes):
    return [rng.integers(0, 256, (h, w, 3), dtype=np.uint8) for h, w in shapes]


def test_voc_palette_known_colours():
    pal = voc_palette()
    assert pal.shape == (256, 3)
    assert pal[0].tolist() == [0, 0, 0] and pal[1].tolist() == [128, 0, 0] and pal[2].tolist() == [0, 128, 0]
    assert pal[15].tolist() == [192, 128, 128] and pal[255].tolist() == [224, 224, 192]


def test_semseg_predictor_sizes_and_stretch_equals_direct_forward():
    torch.manual_seed(0)
    rng = np.random.default_rng(0)
    model = build_model("fcn_resnet18", 5)
    crop = 65
    pred = SemSegPredictor(model, "cpu", crop=crop, fit="stretch", batch=2)
    imgs = _images(rng, [(crop, crop), (40, 70), (90, 33)])
    out = pred.predict(imgs)
    assert [o.shape for o in out] == [(crop, crop), (40, 70), (90, 33)]
    assert all(o.dtype == np.uint8 and o.max() < 5 for o in out)
    # an image that already has the network's size: the gather is the identity, the result the plain forward's
    x = (torch.from_numpy(imgs[0]).permute(2, 0, 1).float() - torch.tensor(IMAGENET_MEAN).view(3, 1, 1)) \
        / torch.tensor(IMAGENET_STD).view(3, 1, 1)
    with torch.no_grad():
        ref = model.eval()(x[None])
    ref = ref[0] if isinstance(ref, (tuple, list)) else ref
    lg = pred.logits(imgs[:1])[0]
    assert torch.allclose(lg, ref[0].float(), atol=1e-4, rtol=1e-4)
    assert (out[0] == ref[0].argmax(0).numpy()).mean() > 0.999


def test_semseg_letterbox_keeps_aspect_and_ignores_padding_side():
    torch.manual_seed(1)
    rng = np.random.default_rng(1)
    model = build_model("fcn_resnet18", 3)
    pred = SemSegPredictor(model, "cpu", crop=64, fit="letterbox", batch=4)
    assert pred._extent(100, 50) == (64, 32) and pred._extent(30, 60) == (32, 64) and pred._extent(64, 64) == (64, 64)
    img = _images(rng, [(100, 50)])[0]
    a = pred.logits([img])[0]
    assert a.shape == (3, 100, 50) and torch.isfinite(a).all()
    # a batch neighbour of another size changes the padded canvas, not the image's own result
    b = pred.logits([img, _images(rng, [(120, 140)])[0]])[0]
    assert torch.allclose(a, b, atol=1e-4, rtol=1e-4)


def test_instance_predictor_matches_the_validation_chain():
    """Points taken from a mask give the crop the validation chain makes from that mask: same image channels, a
    guidance channel that differs only by the rounding of the points, and a probability map of the image's size."""
    torch.manual_seed(2)
    rng = np.random.default_rng(2)
    h, w, crop = 90, 120, 64
    img = _images(rng, [(h, w)])[0]
    yy, xx = np.mgrid[:h, :w]
    mask = ((((yy - 45) / 20.0) ** 2 + ((xx - 60) / 35.0) ** 2) <= 1.0).astype(np.uint8)
    pts = helpers.extreme_points_fixed(mask)                  # [4, 2] (x, y)
    model = build_model("fcn_resnet18", 1, in_channels=4)
    pred = InstancePredictor(model, "cpu", crop=crop, relax=10, zero_pad=True)
    x, cb, sizes = pred.network_input([img], [pts])
    assert x.shape == (1, 4, crop, crop) and sizes.tolist() == [[h, w]]
    ident = torch.tensor([[1., 0, 0, 0, 1, 0]])
    ref, _, ref_bb = instance_augment(torch.from_numpy(img)[None], torch.from_numpy(mask)[None],
                                      torch.zeros(1, dtype=torch.int32), torch.tensor([[h, w]], dtype=torch.int32),
                                      ident, (crop, crop), 10, True, random_points=False, out_dtype=torch.float32)
    assert cb[0].tolist() == ref_bb[0].tolist()
    assert torch.allclose(x[:, :3], ref[:, :3], atol=1e-4)
    assert float((x[:, 3] - ref[:, 3]).abs().mean()) < 12.0    # 0..255 scale; the points move by at most a pixel
    prob = pred.predict([img, img], [pts, pts])
    assert len(prob) == 2 and prob[0].shape == (h, w) and prob[0].dtype == np.float32
    assert np.allclose(prob[0], prob[1]) and 0.0 <= prob[0].min() and prob[0].max() <= 1.0
    # nothing is predicted outside the relaxed box
    x0, y0, x1, y1 = cb[0]
    outside = np.ones((h, w), bool)
    outside[max(y0, 0):y1 + 1, max(x0, 0):x1 + 1] = False
    assert prob[0][outside].max(initial=0.0) == 0.0


def test_instance_predictor_refuses_points_outside_the_image():
    model = build_model("fcn_resnet18", 1, in_channels=4)
    pred = InstancePredictor(model, "cpu", crop=32, relax=5)
    img = np.zeros((40, 50, 3), np.uint8)
    try:
        pred.predict([img], [[[5, 5], [60, 5], [20, 2], [20, 30]]])
    except ValueError as e:
        assert "outside" in str(e)
    else:
        raise AssertionError("points beyond the image were accepted")


def test_predict_cli_writes_paletted_maps(tmp_path):
    import predict
    from PIL import Image
    rng = np.random.default_rng(3)
    src = tmp_path / "in"
    src.mkdir()
    for i, (h, w) in enumerate([(50, 70), (64, 40)]):
        Image.fromarray(rng.integers(0, 256, (h, w, 3), dtype=np.uint8)).save(src / f"im{i}.png")
    out = tmp_path / "out"
    rec = predict.main(["--cpu", "--model", "fcn_resnet18", "--crop", "65", "--input", str(src), "--output", str(out),
                        "--batch", "2"])
    assert rec["images"] == 2 and rec["device"] == "cpu" and rec["graph"] is False
    a = Image.open(out / "im0.png")
    assert a.mode == "P" and a.size == (70, 50) and np.asarray(a).max() < 21
    assert Image.open(out / "im1.png").size == (40, 64)
    # the instance task from the command line
    rec = predict.main(["--cpu", "--task", "instance", "--model", "fcn_resnet18", "--crop", "64",
                        "--input", str(src / "im0.png"), "--points", "10,25 60,25 35,5 35,45", "--output", str(out / "inst")])
    m = np.asarray(Image.open(out / "inst" / "im0.png"))
    assert m.shape == (50, 70) and set(np.unique(m).tolist()) <= {0, 255}
    # synthetic rate run prints a steady rate and writes nothing
    rec = predict.main(["--cpu", "--model", "fcn_resnet18", "--crop", "65", "--synthetic", "3", "--synthetic-size", "40x60",
                        "--no-write"])
    assert rec["steady"] and rec["forward_img_per_s"] > 0 and rec["output"] is None
    json.dumps(rec)


def test_predictor_over_a_training_engines_model_leaves_the_arena_intact():
    from distributedpytorch_amd.ops import SoftmaxCrossEntropy
    from distributedpytorch_amd.parallel import TrainEngine
    torch.manual_seed(4)
    engine = TrainEngine(build_model("fcn_resnet18", 5), SoftmaxCrossEntropy(255), lr=0.05, momentum=0.9, weight_decay=0.0,
                         device=torch.device("cpu"))
    ptrs = [p.data_ptr() for p in engine.model.parameters()]
    pred = SemSegPredictor(engine.model, "cpu", crop=65, batch=2)
    img = np.random.default_rng(4).integers(0, 256, (50, 60, 3), dtype=np.uint8)
    before = pred.logits([img])[0].clone()
    assert ptrs == [p.data_ptr() for p in engine.model.parameters()]
    engine.model.train()
    x, y = torch.randn(2, 3, 65, 65), torch.randint(0, 5, (2, 65, 65))
    for _ in range(2):
        engine.train_step(x, y)
    engine.model.eval()
    after = pred.logits([img])[0]
    assert not torch.allclose(before, after)                      # the predictor sees the trained weights and statistics


def test_predict_cli_two_ranks_deal_the_images(tmp_path):
    """`predict.py --gpus 2` (here: two CPU ranks over gloo, started by the script itself): every image is written
    exactly once, the maps equal the one-process run's, rank 0 prints the job's line."""
    import subprocess
    from PIL import Image
    rng = np.random.default_rng(5)
    src = tmp_path / "in"
    src.mkdir()
    for i, (h, w) in enumerate([(50, 70), (64, 40), (33, 91), (80, 80), (45, 52)]):
        Image.fromarray(rng.integers(0, 256, (h, w, 3), dtype=np.uint8)).save(src / f"im{i}.png")
    base = [sys.executable, os.path.join(ROOT, "predict.py"), "--cpu", "--model", "fcn_resnet18", "--crop", "65",
            "--input", str(src), "--batch", "2"]
    env = dict(os.environ, OMP_NUM_THREADS="2")
    one = subprocess.run(base + ["--output", str(tmp_path / "one")], capture_output=True, text=True, timeout=300, env=env)
    assert one.returncode == 0, one.stderr[-2000:]
    two = subprocess.run(base + ["--output", str(tmp_path / "two"), "--gpus", "2"], capture_output=True, text=True,
                         timeout=300, env=env)
    assert two.returncode == 0, two.stderr[-2000:]
    lines = [l for l in two.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    rec = json.loads(lines[0])
    assert rec["images"] == 5 and rec["n_gpus"] == 2
    assert sorted(os.listdir(tmp_path / "two")) == sorted(os.listdir(tmp_path / "one")) == [f"im{i}.png" for i in range(5)]
    for i in range(5):
        a = np.asarray(Image.open(tmp_path / "one" / f"im{i}.png"))
        b = np.asarray(Image.open(tmp_path / "two" / f"im{i}.png"))
        assert (a == b).mean() > 0.999
