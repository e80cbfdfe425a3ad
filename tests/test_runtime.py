"""Round-2 runtime pieces on CPU: self-launch, content-hashed build, unpadded evaluation sampler, tail
bucket, device data path (torch reference path) vs the CPU transforms, SBD + CombineDBs, resume semantics,
debug-mode batch sanity.  Reference behaviour cited per test."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


# ------------------------------------------------------------------------------ launcher
def test_self_launch_starts_n_ranks_and_propagates_failure(tmp_path):
    from distributedpytorch_amd.parallel.launch import launch_ranks
    ok = tmp_path / "ok.py"
    ok.write_text("import os,sys\nopen(sys.argv[1]+'/r'+os.environ['RANK'],'w').write("
                  "os.environ['WORLD_SIZE']+' '+os.environ['LOCAL_RANK']+' '+os.environ['MASTER_PORT']+' '"
                  "+os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY','unset'))\n")
    assert launch_ranks(3, [sys.executable, str(ok), str(tmp_path)]) == 0
    got = [(tmp_path / f"r{r}").read_text().split() for r in range(3)]
    assert [g[0] for g in got] == ["3"] * 3 and [g[1] for g in got] == ["0", "1", "2"]
    assert len({g[2] for g in got}) == 1 and all(g[3] == "0" for g in got)      # one port; dmabuf IPC exported
    bad = tmp_path / "bad.py"
    bad.write_text("import os,sys,time\nif os.environ['RANK']=='1': sys.exit(7)\ntime.sleep(60)\n")
    assert launch_ranks(2, [sys.executable, str(bad)]) == 7          # rank 1 dies -> rank 0 is stopped, status kept


def test_launcher_counts_gpus_from_the_kfd_topology(tmp_path, monkeypatch):
    """the parent counted devices with torch.cuda.device_count(), which starts the HIP runtime when
    amdsmi is absent.  It reads the KFD topology from sysfs instead: nodes with simd_count > 0 are GPUs."""
    from distributedpytorch_amd.parallel import launch
    for i, simd in enumerate((0, 0, 1024, 1024, 1024)):
        d = tmp_path / str(i)
        d.mkdir()
        (d / "properties").write_text(f"cpu_cores_count {64 if simd == 0 else 0}\nsimd_count {simd}\ngfx_target_version 90500\n")
    assert launch.kfd_gpu_count(str(tmp_path)) == 3
    assert launch.kfd_gpu_count(str(tmp_path / "nope")) is None
    for var in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        monkeypatch.delenv(var, raising=False)
    assert launch._mask_count(3) == 3
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "0,2")
    assert launch._mask_count(3) == 2
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "0,7,1")                  # the runtime stops at the first bad index
    assert launch._mask_count(3) == 1
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "")
    assert launch._mask_count(3) == 0


def test_bench_self_launch_cpu_reports_two_ranks():
    """`python bench.py --gpus 2` (no torchrun) prints ONE JSON line with n_gpus = 2 — the driver's contract and
    the reference's one-command-many-GPUs behaviour (`/root/reference/train_pascal.py:92`)."""
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--cpu", "--gpus", "2", "--model",
                        "fcn_resnet18", "--crop", "64", "--batch", "2", "--steps", "2", "--warmup", "1"],
                       capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["config"]["parallelism"] == "dp2" and rec["config"]["global_batch"] == 4
    assert rec["comm"]["backend"] == "gloo" and len(rec["comm"]["devices"]) == 2
    assert "comm_wait_ms" in rec["comm"] and rec["steps"] == 2 and rec["warmup"] == 1


def test_bench_refuses_mislabelled_world(tmp_path):
    env = dict(os.environ, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--cpu", "--gpus", "2", "--model",
                        "fcn_resnet18", "--crop", "64", "--batch", "2", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode != 0 and "WORLD_SIZE=1" in (r.stderr + r.stdout)


# ------------------------------------------------------------------------------ build provenance
def test_build_hash_is_content_based(tmp_path):
    from distributedpytorch_amd.csrc import build as b
    h = b.source_hash()
    assert len(h) == 32 and h == b.source_hash()
    lib = tmp_path / "lib.so"
    lib.write_bytes(b"\x7fELF....DP_BUILD_HASH=" + h.encode() + b"\0....")
    assert b.embedded_hash(str(lib)) == h
    assert b.embedded_hash(str(tmp_path / "missing.so")) == ""
    if os.path.isfile(b.HIP_LIB):
        assert b.embedded_hash(b.HIP_LIB) == h, "in-tree libdpamd_hip.so is stale: run csrc/build.py"


# ------------------------------------------------------------------------------ sampler / arena
def test_eval_sampler_visits_every_sample_once():
    from distributedpytorch_amd.parallel import DistributedSampler
    n, world = 11, 4
    shards = [list(DistributedSampler(n, world, r, shuffle=False, drop_last=False, pad=False)) for r in range(world)]
    flat = sorted(i for s in shards for i in s)
    assert flat == list(range(n))                              # no duplicates, nothing dropped
    assert max(map(len, shards)) - min(map(len, shards)) <= 1
    assert [len(DistributedSampler(n, world, r, shuffle=False, drop_last=False, pad=False)) for r in range(world)] \
        == [len(s) for s in shards]
    padded = [list(DistributedSampler(n, world, r, shuffle=False, drop_last=False)) for r in range(world)]
    assert all(len(s) == 3 for s in padded)                    # training-style: equal counts via wrap-around


def test_arena_tail_bucket_is_small():
    from distributedpytorch_amd.models import build_model
    from distributedpytorch_amd.parallel import ParamArena
    m = build_model("fcn_resnet18", 21)
    a0 = ParamArena([list(m.named_parameters())], bucket_bytes=16 << 20)
    m2 = build_model("fcn_resnet18", 21)
    a1 = ParamArena([list(m2.named_parameters())], bucket_bytes=16 << 20, tail_bytes=1 << 20)
    assert len(a1.buckets) == len(a0.buckets) + 1
    last = a1.buckets[-1]
    assert (last.end - last.start) * 4 <= (1 << 20) and len(last.slots) >= 1
    # buckets tile the arena, slots know their bucket
    assert a1.buckets[0].start == 0 and a1.buckets[-1].end == a1.total
    for b0, b1 in zip(a1.buckets, a1.buckets[1:]):
        assert b0.end == b1.start
    for bk in a1.buckets:
        assert all(a1.slots[si].bucket == bk.index for si in bk.slots)


# ------------------------------------------------------------------------------ device data path (reference impl)
def _scene(h=72, w=88, seed=0):
    from distributedpytorch_amd.data.pascal import _synthetic_scene
    return _synthetic_scene(np.random.default_rng(seed), h, w)


@pytest.mark.parametrize("interp", ["bilinear", "bicubic"])
def test_affine_augment2_matches_cpu_warp_on_a_fixed_matrix(interp):
    """CPU-vs-device augmentation equivalence on a FIXED matrix: ``ops.affine_augment2`` against the C++/NumPy
    ``warp_affine`` the CPU transforms use (`/root/reference/custom_transforms.py:116-126`: cv2.warpAffine)."""
    from distributedpytorch_amd import ops
    from distributedpytorch_amd.data import imgproc
    img, cls, _ = _scene()
    h, w = cls.shape
    fwd = imgproc.get_rotation_matrix_2d((w / 2, h / 2), 13.0, 1.17)          # source -> canvas
    inv = imgproc.invert_affine(fwd)
    mats = torch.tensor(inv.reshape(1, 6), dtype=torch.float32)
    flag = imgproc.INTER_CUBIC if interp == "bicubic" else imgproc.INTER_LINEAR
    want = imgproc.warp_affine(img.astype(np.float32), fwd, (w, h), flag, 0.0)
    want_lab = imgproc.warp_affine(cls.astype(np.float32), fwd, (w, h), imgproc.INTER_NEAREST, 255.0)
    x, y = ops.affine_augment2(torch.from_numpy(img)[None], torch.from_numpy(cls)[None], mats, (h, w),
                               interp=interp, mean=(0, 0, 0), std=(1, 1, 1), border=(0, 0, 0),
                               out_dtype=torch.float32)
    got = x[0].permute(1, 2, 0).numpy()
    assert np.abs(got - want).max() < 0.05 * 255 and np.abs(got - want).mean() < 0.02      # fp32 rounding only
    assert (y[0].numpy() != want_lab.astype(np.uint8)).mean() < 2e-3                       # nearest ties at .5


def test_affine_augment2_slots_sizes_and_clip():
    from distributedpytorch_amd import ops
    store = torch.zeros(3, 40, 48, 3, dtype=torch.uint8)
    lab = torch.full((3, 40, 48), 255, dtype=torch.uint8)
    store[2, :20, :30] = 200
    lab[2, :20, :30] = 7
    ident = torch.tensor([[1.0, 0, 0, 0, 1, 0]])
    x, y = ops.affine_augment2(store, lab, ident, (40, 48), src_idx=torch.tensor([2], dtype=torch.int32),
                               sizes=torch.tensor([[20, 30]], dtype=torch.int32), mean=(0, 0, 0), std=(1, 1, 1),
                               border=(9, 9, 9), out_dtype=torch.float32)
    assert torch.all(y[0, :20, :30] == 7) and torch.all(y[0, 20:] == 255) and torch.all(y[0, :, 30:] == 255)
    assert torch.all(x[0, :, :19, :29] == 200) and torch.all(x[0, :, 21:] == 9)        # beyond the valid size: border
    clip = torch.tensor([[1.0, 0, 0, 0, 1, 0, 10, 8]])                                # canvas 10 wide, 8 high
    x2, y2 = ops.affine_augment2(store, lab, ident, (40, 48), src_idx=torch.tensor([2], dtype=torch.int32),
                                 sizes=torch.tensor([[20, 30]], dtype=torch.int32), clip=clip, lab_fill=0,
                                 mean=(0, 0, 0), std=(1, 1, 1), border=(0, 0, 0), out_dtype=torch.float32)
    assert torch.all(y2[0, :8, :10] == 7) and torch.all(y2[0, 8:] == 0) and torch.all(y2[0, :, 10:] == 0)
    assert torch.all(x2[0, :, :8, :10] == 200) and torch.all(x2[0, :, 8:] == 0)


def test_instance_augment_matches_the_cpu_transform_chain():
    """Device chain (one composed bicubic gather) vs the reference's CPU chain
    (`/root/reference/train_pascal.py:123-134`) on the same object, no flip / rotation / scale: crop geometry,
    guidance map and value ranges agree (one resampling instead of two: interiors match, edges differ slightly)."""
    from distributedpytorch_amd import ops
    from distributedpytorch_amd.data import transforms as tr
    img, cls, inst = _scene(96, 120, seed=3)
    gt = ((inst == 1)).astype(np.float32)
    assert gt.sum() > 50
    sample = {"image": img.astype(np.float32), "gt": gt.copy(), "void_pixels": np.zeros_like(gt)}
    chain = tr.Compose([tr.CropFromMaskStatic(crop_elems=("image", "gt"), relax=10, zero_pad=True),
                        tr.FixedResize(resolutions={"crop_image": (64, 64), "crop_gt": (64, 64)}),
                        tr.NEllipseWithGaussians(alpha=0.6, is_val=True),
                        tr.ConcatInputs(elems=("crop_image", "nellipseWithGaussians")), tr.ToTensor()])
    want = chain(sample)
    ident = torch.tensor([[1.0, 0, 0, 0, 1, 0]])
    x, y, bbox = ops.instance_augment(torch.from_numpy(img)[None], torch.from_numpy(gt.astype(np.uint8))[None],
                                      torch.tensor([0], dtype=torch.int32),
                                      torch.tensor([[96, 120]], dtype=torch.int32), ident, (64, 64), relax=10,
                                      zero_pad=True, random_points=False, out_dtype=torch.float32)
    assert x.shape == (1, 4, 64, 64) and y.shape == (1, 1, 64, 64)
    ys, xs = np.where(gt > 0)
    assert bbox[0].tolist() == [xs.min() - 10, ys.min() - 10, xs.max() + 10, ys.max() + 10]
    wy = want["crop_gt"][0].numpy()
    iou = ((y[0, 0].numpy() > 0) & (wy > 0)).sum() / max(((y[0, 0].numpy() > 0) | (wy > 0)).sum(), 1)
    assert iou > 0.97
    assert set(np.unique(y.numpy()).tolist()) <= {0.0, 1.0}
    wi = want["concat"][:3].numpy()
    assert np.abs(x[0, :3].numpy() - wi).mean() < 4.0                      # 0..255 scale
    g, wg = x[0, 3].numpy(), want["concat"][3].numpy()
    assert g.min() >= 0 and g.max() <= 255.001 and abs(g.max() - 255) < 1e-2
    assert np.abs(g - wg).mean() < 6.0                                     # extreme points may differ by a pixel
    # the reference's in-loop asserts (train_pascal.py:188-190) hold on the device batch
    from distributedpytorch_amd.utils.sanity import BatchSanity
    chk = BatchSanity("instance", every=1)
    chk.check(x, y)


def test_extreme_points_and_bbox_reference_semantics():
    from distributedpytorch_amd import ops
    from distributedpytorch_amd.data import helpers
    _, _, inst = _scene(64, 64, seed=5)
    m = (inst == 1).astype(np.uint8)
    pts = ops.extreme_points(torch.from_numpy(m)[None], random=False)[0].numpy()
    ref = helpers.extreme_points_fixed(m.astype(np.float32), 0)
    assert np.array_equal(pts, ref)
    ys, xs = np.where(m > 0)
    rp = ops.extreme_points(torch.from_numpy(m)[None], random=True, seed=3)[0].numpy()
    assert rp[0, 0] == xs.min() and rp[1, 0] == xs.max() and rp[2, 1] == ys.min() and rp[3, 1] == ys.max()
    assert all(m[p[1], p[0]] for p in rp)
    empty = ops.extreme_points(torch.zeros(1, 8, 8, dtype=torch.uint8), random=False)
    assert (empty == -1).all()
    bb = ops.mask_bbox(torch.from_numpy(m)[None], torch.tensor([[1.0, 0, 0, 0, 1, 0]]))[0].tolist()
    assert bb == list(helpers.get_bbox(m))


def test_gpu_batch_loader_cpu_semseg_and_instance():
    from distributedpytorch_amd.data import SyntheticVOC
    from distributedpytorch_amd.data.gpu_pipeline import GpuBatchLoader
    from distributedpytorch_amd.parallel import DistributedSampler
    dev = torch.device("cpu")
    for task, mode, ch in (("semseg", "semantic", 3), ("instance", "instance", 4)):
        db = SyntheticVOC(6, (40, 48), None, mode, seed=1)
        for loader_mode in ("cache", "stream"):
            ld = GpuBatchLoader(db, DistributedSampler(len(db), 1, 0, shuffle=True, seed=0), 2, dev, 32, task,
                                workers=0, mode=loader_mode, rots=(-20, 20) if task == "instance" else (0, 0),
                                scales=(.75, 1.25) if task == "instance" else (0.5, 2.0))
            ld.set_epoch(0)
            batches = list(ld)
            assert len(batches) == len(ld) == 3
            x, y = batches[0]
            assert x.shape == (2, ch, 32, 32) and torch.isfinite(x).all()
            if task == "semseg":
                assert y.dtype == torch.uint8 and y.shape == (2, 32, 32)
                assert set(np.unique(y.numpy()).tolist()) <= set(range(21)) | {255}
            else:
                assert y.shape == (2, 1, 32, 32) and set(np.unique(y.numpy()).tolist()) <= {0.0, 1.0}
                assert x[:, :3].min() >= 0 and x[:, :3].max() <= 255
    # two ranks draw different augmentation parameters (ADVICE: correlated augmentation across ranks)
    db = SyntheticVOC(4, (40, 48), None, "semantic", seed=1)
    outs = []
    for rank in (0, 1):
        ld = GpuBatchLoader(db, DistributedSampler(len(db), 1, 0, shuffle=False), 2, dev, 32, "semseg", workers=0,
                            seed=0, rank=rank)
        outs.append(next(iter(ld))[0])
    assert not torch.equal(outs[0], outs[1])


def test_cpu_loader_workers_are_seeded_per_rank():
    """ADVICE (round 1): every rank's worker w drew the same ScaleNRotate parameters.  The worker_init_fn mixes the
    rank into NumPy's seed."""
    import train_pascal as tp
    from torch.utils.data import DataLoader, Dataset

    class Draw(Dataset):
        def __len__(self):
            return 4

        def __getitem__(self, i):
            return torch.tensor(np.random.random())

    def draws(rank):
        gen = torch.Generator().manual_seed(rank)
        dl = DataLoader(Draw(), batch_size=1, num_workers=2, generator=gen, worker_init_fn=tp._seed_worker(rank))
        return [float(v) for v in dl]
    a, b = draws(1000), draws(1001)
    assert a != b and len(set(a)) == 4


# ------------------------------------------------------------------------------ SBD + CombineDBs
def test_sbd_dataset_and_combine_with_voc(tmp_path):
    from distributedpytorch_amd.data import CombineDBs, VOCSegmentation
    from distributedpytorch_amd.data.pascal import write_synthetic_voc_tree
    from distributedpytorch_amd.data.sbd import SBDSegmentation, write_synthetic_sbd_tree
    voc_root = write_synthetic_voc_tree(str(tmp_path / "voc"), n_train=3, n_val=2, size=(64, 80))
    voc_train = VOCSegmentation(voc_root, "train", check_integrity=False)
    voc_val = VOCSegmentation(voc_root, "val", check_integrity=False)
    sbd_root = write_synthetic_sbd_tree(str(tmp_path / "sbd"), n_train=3, n_val=1, size=(64, 80),
                                        shared_ids=[voc_val.im_ids[0]])
    sbd = SBDSegmentation(sbd_root, ["train", "val"])
    assert len(sbd) > 0 and str(sbd) == "SBD(split=['train', 'val'],area_thres=0)"
    s = sbd[0]
    assert set(s) == {"image", "gt", "void_pixels", "meta"} and s["image"].dtype == np.float32
    assert set(np.unique(s["gt"]).tolist()) <= {0.0, 1.0} and s["gt"].sum() > 0
    sem = SBDSegmentation(sbd_root, "train", semantic=True)
    assert sem[0]["gt"].max() <= 20 and len(sem) == 3
    combo = CombineDBs([voc_train, sbd], excluded=[voc_val])      # train_pascal.py:152
    ids = {combo.dataloaders[e["db_ii"]].im_ids[combo.dataloaders[e["db_ii"]].obj_list[e["cat_ii"]][0]]
           for e in combo.cat_list}
    assert voc_val.im_ids[0] not in ids and len(combo) < len(voc_train) + len(sbd)
    assert combo[0]["image"].ndim == 3


# ------------------------------------------------------------------------------ resume / checkpoint
def test_resume_restores_best_val_and_rng(tmp_path):
    """The reference keeps ``best_val`` as the save gate across epochs (`/root/reference/train_pascal.py:177`,
    `:301-304`); a resumed run must not restart it at 0 and overwrite a better ``_best.pth``."""
    import train_pascal as tp
    common = ["--synthetic", "--device", "cpu", "--model", "fcn_resnet18", "--crop", "48", "--trainBatch", "2",
              "--synthetic-len", "8", "--workers", "0", "--save-dir", str(tmp_path), "--log-every", "0",
              "--snapshot", "1", "--max-iters", "2"]
    tp.main(common + ["--nEpochs", "2"])
    side = torch.load(str(tmp_path / "run_0" / "models" / "fcn_resnet18_epoch-1.pth.state"), weights_only=True)
    assert {"rng_cpu", "np_rng", "dropout_calls", "optimizer", "extra"} <= set(side)
    side["extra"]["best_val"] = 0.99              # pretend an excellent model was saved before the interruption
    torch.save(side, str(tmp_path / "run_0" / "models" / "fcn_resnet18_epoch-1.pth.state"))
    best_file = tmp_path / "run_0" / "models" / "fcn_resnet18_best.pth"
    stamp = best_file.stat().st_mtime_ns
    best = tp.main(common + ["--nEpochs", "3", "--resume_epoch", "2"])
    assert best == pytest.approx(0.99)            # carried over; the worse epoch-2 model did not replace the best
    assert best_file.stat().st_mtime_ns == stamp
    with pytest.raises(FileNotFoundError):
        tp.main(common + ["--nEpochs", "9", "--resume_epoch", "7"])


def test_rng_state_roundtrip():
    from distributedpytorch_amd.ops import batchnorm as bn
    from distributedpytorch_amd.utils import checkpoint as ck
    torch.manual_seed(5)
    np.random.seed(11)
    bn._drop_calls = 42
    st = ck.capture_rng_state()
    a = (torch.rand(3), np.random.random(3))
    bn._drop_calls = 99
    ck.restore_rng_state(st)
    b = (torch.rand(3), np.random.random(3))
    assert torch.equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and bn._drop_calls == 42


def test_nonfinite_guard_restores_last_checkpoint(tmp_path, monkeypatch):
    """non-finite loss -> every rank raises together; ``--on-nonfinite restore`` reloads the last
    checkpoint of the run and continues."""
    import train_pascal as tp
    from distributedpytorch_amd.parallel import TrainEngine
    common = ["--synthetic", "--device", "cpu", "--model", "fcn_resnet18", "--crop", "48", "--trainBatch", "2",
              "--synthetic-len", "8", "--workers", "0", "--save-dir", str(tmp_path), "--log-every", "0",
              "--snapshot", "1", "--finite-every", "1", "--useTest", "0"]
    orig = TrainEngine.train_step
    state = {"n": 0}

    def poisoned(self, x, y):
        loss = orig(self, x, y)
        state["n"] += 1
        return loss * float("nan") if state["n"] == 10 else loss      # epoch 2 of a 4-step epoch
    monkeypatch.setattr(TrainEngine, "train_step", poisoned)
    with pytest.raises(FloatingPointError):
        tp.main(common + ["--nEpochs", "3"])
    state["n"] = 0
    tp.main(common + ["--nEpochs", "3", "--on-nonfinite", "restore", "--save-dir", str(tmp_path / "b")])
    log = (tmp_path / "b" / "run_0" / "log.jsonl").read_text()
    assert '"kind": "restore"' in log and log.count('"kind": "epoch"') == 3


def test_batch_sanity_flags_bad_batches():
    """The reference's per-batch asserts (`/root/reference/train_pascal.py:188-190`) as device-side predicates."""
    from distributedpytorch_amd.utils.sanity import BatchSanity
    x = torch.rand(2, 4, 16, 16) * 255
    y = (torch.rand(2, 1, 16, 16) > 0.5).float()
    BatchSanity("instance", every=1).check(x, y)
    for bad_x, bad_y, word in ((x * 2, y, "RGB"), (x, y * 3, "ground truth"), (torch.zeros_like(x), y, "distinct")):
        with pytest.raises(AssertionError, match=word):
            BatchSanity("instance", every=1).check(bad_x, bad_y)
    lab = torch.randint(0, 21, (2, 16, 16))
    lab[:, :2] = 255
    BatchSanity("semseg", every=1, normalised=True).check(torch.randn(2, 3, 16, 16), lab)
    with pytest.raises(AssertionError, match="labels"):
        BatchSanity("semseg", every=1, normalised=True).check(torch.randn(2, 3, 16, 16), lab + 30)


def test_validation_figure_written(tmp_path):
    """4-panel validation PNG (`/root/reference/train_pascal.py:257-278`) lands in the run directory."""
    import train_pascal as tp
    tp.main(["--task", "instance", "--synthetic", "--device", "cpu", "--model", "danet_resnet18", "--crop", "64",
             "--nEpochs", "1", "--trainBatch", "2", "--synthetic-len", "4", "--workers", "0", "--save-dir",
             str(tmp_path), "--log-every", "0", "--max-iters", "1"])
    png = tmp_path / "run_0" / "figures" / "val_epoch-0.png"
    assert png.is_file() and png.stat().st_size > 2000


# ------------------------------------------------------------------------------ validation on the device path
def test_device_validation_matches_the_host_validation(tmp_path):
    """validation ran on the host (CPU transforms in DataLoader workers, every prediction copied back,
    per-object NumPy Jaccard).  The device path — GpuBatchLoader(train=False) + ops.paste_jaccard / the device
    confusion matrix — must give the same numbers.  Here both run on CPU tensors (the torch formulation of every
    device op); tests/test_gpu_runtime.py repeats the metric kernel against this formulation on the GPU."""
    import copy
    import types

    import train_pascal as tp
    from distributedpytorch_amd.data import SyntheticVOC
    from distributedpytorch_amd.data.gpu_pipeline import GpuBatchLoader
    from distributedpytorch_amd.models import build_model
    from distributedpytorch_amd.parallel import DistributedSampler, TrainEngine
    from distributedpytorch_amd.utils import SegmentationMultiLosses
    from torch.utils.data import DataLoader
    dev = torch.device("cpu")
    # ---- instance task: per-object Jaccard at .3/.5/.8 ----
    args = types.SimpleNamespace(task="instance", crop=96, zero_pad_crop=1, max_iters=0)
    db = SyntheticVOC(6, (120, 150), tp.build_transforms(args, False), "instance", seed=3)
    raw = copy.copy(db)
    raw.transform = None

    class FromGuidance(torch.nn.Module):
        """A "network" with a graded, object-shaped prediction (an untrained one saturates on 0..255 inputs): the logit
        is an affine function of the n-ellipse guidance channel mixed with the image, so the three thresholds cut
        three different masks and every stage of the input chain matters."""

        def __init__(self):
            super().__init__()
            self.k = torch.nn.Parameter(torch.tensor(1.0 / 30.0))

        def forward(self, x):
            x = x.float()
            return (x[:, 3:4] - 110.0 + 0.1 * (x[:, 0:1] - 128.0)) * self.k

    eng = TrainEngine(FromGuidance(), SegmentationMultiLosses(1), lr=0.0, device=dev)
    sampler = DistributedSampler(len(db), shuffle=False, drop_last=False, pad=False)
    host = tp.validate_instance(eng, DataLoader(db, batch_size=1, sampler=sampler), args)
    ld = GpuBatchLoader(raw, sampler, 4, dev, 96, "instance", workers=0, train=False, relax=50, zero_pad=True)
    assert len(ld) == 2                         # 6 objects in batches of 4: the tail batch is visited
    device = tp.validate_instance_device(eng, ld, args)
    assert device["img_per_s"] > 0
    for a, b in zip(host["jaccard"], device["jaccard"]):
        # the host chain resamples twice (crop, then resize), the device chain once: a few border pixels differ
        assert abs(a - b) < 2e-3, (host, device)
    assert 0.0 < max(device["jaccard"]) <= 1.0 and len({round(v, 3) for v in device["jaccard"]}) == 3
    # ---- semantic task: mIoU from the device confusion matrix ----
    from distributedpytorch_amd import ops
    sargs = types.SimpleNamespace(task="semseg", crop=64, zero_pad_crop=1, max_iters=0)
    sdb = SyntheticVOC(5, (80, 100), tp.build_transforms(sargs, False), "semantic", seed=4)
    sraw = copy.copy(sdb)
    sraw.transform = None
    torch.manual_seed(1)
    seng = TrainEngine(build_model("fcn_resnet18", 21), ops.SoftmaxCrossEntropy(255), lr=0.0, device=dev)
    ssampler = DistributedSampler(len(sdb), shuffle=False, drop_last=False, pad=False)
    h = seng.evaluate(tp._KeyLoader(DataLoader(sdb, batch_size=2, sampler=ssampler), "image", "gt"), 21, 255)
    sld = GpuBatchLoader(sraw, ssampler, 2, dev, 64, "semseg", workers=0, train=False)
    assert len(sld) == 3
    d = seng.evaluate(tp._Ticked(sld), 21, 255)
    assert abs(h["miou"] - d["miou"]) < 0.01 and abs(h["pixel_acc"] - d["pixel_acc"]) < 0.01, (h, d)


def test_paste_jaccard_counts():
    """ops.paste_jaccard on a hand-made case: a confident square prediction pasted back to a box that reaches outside
    the image, void pixels excluded, an empty box."""
    from distributedpytorch_amd.ops.augment import paste_jaccard
    gt = torch.zeros(2, 40, 50, dtype=torch.uint8)
    gt[0, 10:30, 15:35] = 1
    void = torch.zeros_like(gt)
    void[0, 10:12] = 1                                   # two object rows are void
    logits = torch.full((2, 1, 32, 32), -9.0)
    logits[0, 0, 8:24, 8:24] = 9.0                       # the centre quarter of the crop
    bbox = torch.tensor([[5, 0, 44, 39], [0, 0, -1, -1]], dtype=torch.int32)      # 40 x 40 box; none
    sizes = torch.tensor([[40, 50], [40, 50]], dtype=torch.int32)
    c = paste_jaccard(logits, bbox, gt, void, None, sizes, relax=0, thresholds=(0.5,))
    # crop quarter -> box pixels 10..29 in both directions -> image x 15..34, y 10..29: exactly the object
    assert c[0, 0].tolist() == [18 * 20, 18 * 20] and c[1, 0].tolist() == [0, 0]
    c = paste_jaccard(logits, bbox, gt, None, None, sizes, relax=12, thresholds=(0.5, 0.9))
    # relax 12 keeps box pixels 12..27: 16 x 16 predicted, all inside the 20 x 20 object
    assert c[0, 0].tolist() == [256, 400] and c[0, 1].tolist() == [256, 400]


def test_device_forms_of_the_unused_transforms_match_the_cpu_transforms():
    """VERDICT r2 "missing" #3: CreateBBMask, ExtremePoints, AddConfidenceMap and CropFromMask's dynamic relax existed
    as CPU transforms only.  ops.augment has batched tensor forms (any device, no host loop); here on CPU tensors
    against data/transforms.py, sample by sample (tests/test_gpu_runtime.py repeats it on the GPU)."""
    from distributedpytorch_amd import ops
    from distributedpytorch_amd.data import helpers
    from distributedpytorch_amd.data import transforms as tr
    rng = np.random.default_rng(0)
    masks = np.zeros((4, 60, 80), dtype=np.uint8)
    yy, xx = np.mgrid[0:60, 0:80]
    masks[0] = ((yy - 30) ** 2 / 15 ** 2 + (xx - 40) ** 2 / 25 ** 2) <= 1            # ellipse
    masks[1, 10:35, 50:70] = 1                                                       # rectangle
    masks[2] = (np.abs(yy - 20) + np.abs(xx - 25) <= 12)                             # diamond
    # masks[3] stays empty
    dev_m = torch.from_numpy(masks)
    bb = ops.bbox_mask(dev_m).numpy()
    hm = ops.extreme_points_heatmap(dev_m, sigma=10).numpy()
    cm_l = ops.confidence_map(dev_m, "l1l2", tau=1.0).numpy()
    cm_g = ops.confidence_map(dev_m, "gaussian").numpy()
    boxes = ops.augment._identity_bbox(dev_m)
    relax = ops.dynamic_crop_relax(boxes, d=512).numpy()
    for b in range(4):
        gt = masks[b].astype(np.float32)
        assert np.array_equal(bb[b], tr.CreateBBMask()({"gt": gt})["bb_mask"]), b
        ref_hm = tr.ExtremePoints(sigma=10, pert=0, elem="gt", is_val=True)({"gt": gt})["extreme_points"]
        assert np.allclose(hm[b], ref_hm, atol=1e-5), b
        img = rng.random((60, 80, 3)).astype(np.float32)
        for kind, got in (("l1l2", cm_l), ("gaussian", cm_g)):
            ref = tr.AddConfidenceMap(elem="image", hm_type=kind, tau=1, is_val=True)({"image": img, "crop_gt": gt})
            assert np.allclose(got[b], ref["with_hm"][..., 3], atol=2e-2), (b, kind, np.abs(got[b] - ref["with_hm"][..., 3]).max())
        if gt.max() > 0:
            t = tr.CropFromMask(crop_elems=("image", "gt"), zero_pad=True, d=512, is_val=True)
            t({"image": img, "gt": gt})
            assert int(relax[b]) == t.crop_relax, (b, int(relax[b]), t.crop_relax)
            x0, y0, x1, y1 = helpers.get_bbox(gt)
            assert boxes[b].tolist() == [x0, y0, x1, y1]
            grown = ops.expand_bbox(boxes, torch.from_numpy(relax))[b].tolist()
            assert grown == [x0 - t.crop_relax, y0 - t.crop_relax, x1 + t.crop_relax, y1 + t.crop_relax]
    assert ops.expand_bbox(boxes, torch.from_numpy(relax))[3, 2] < ops.expand_bbox(boxes, torch.from_numpy(relax))[3, 0]


def test_hostcopy_returns_independent_contiguous_tensors():
    """utils.hostcopy.to_host (the checkpoint's device -> host path; the device branch runs in
    tests/test_gpu_runtime.py): values and logical shapes kept, channels-last inputs come back contiguous, every
    result owns its storage."""
    from distributedpytorch_amd.utils.hostcopy import to_host
    a = torch.randn(4, 3, 2, 2).contiguous(memory_format=torch.channels_last)
    b = torch.arange(5)
    c = torch.randn(7)[::2]
    out = to_host([a, b, c])
    for src, dst in zip((a, b, c), out):
        assert dst.is_contiguous() and dst.shape == src.shape and dst.dtype == src.dtype and torch.equal(dst, src)
        assert dst.untyped_storage().data_ptr() != src.untyped_storage().data_ptr()
    assert len({t.untyped_storage().data_ptr() for t in out}) == 3
