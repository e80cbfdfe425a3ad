"""One-shot peer-to-peer all-reduce (csrc/p2p_allreduce.hip, parallel/p2p.py) on the GPU.

The ranks are separate PROCESSES that map each other's receive buffers through HIP IPC; on a 1-GPU box
they share device 0 (the kernels still exchange tagged granules through each other's buffers and must
run concurrently), on a multi-GPU node each takes its own device."""
import os
import sys
import socket
import tempfile

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, sync_bn_case):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      HSA_ENABLE_IPC_MODE_LEGACY="0")
    torch.cuda.set_device(rank % torch.cuda.device_count())
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from distributedpytorch_amd.parallel.p2p import OneShotAllReduce
    ar = OneShotAllReduce(None)
    try:
        for it, n in enumerate([1, 7, 513, 2048, 4097, 8192] * 3):
            base = torch.arange(n, device="cuda", dtype=torch.float32) * 0.25 + it
            t = base * (rank + 1)
            ar.all_reduce_(t)
            assert torch.equal(t, base * sum(r + 1 for r in range(world))), (n, it)
        g = torch.Generator(device="cuda").manual_seed(5 + rank)
        t = torch.randn(4096, device="cuda", generator=g)
        mine = t.clone()
        ar.all_reduce_(t)
        # (gloo collectives of the checks run on host tensors: its device-tensor path stalled now and then with two
        # processes on one GPU)
        mine = mine.cpu()
        parts = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(parts, mine)
        want = parts[0].clone()
        for p in parts[1:]:
            want += p                                  # rank order, as the kernel sums
        assert torch.equal(t.cpu(), want)
        ref = t.cpu()
        dist.broadcast(ref, src=0)
        assert torch.equal(ref, t.cpu()), "ranks must hold bit-identical results"
        if sync_bn_case:
            # SyncBN through the one-shot kernel == BN over the concatenated batch
            from distributedpytorch_amd import ops
            from distributedpytorch_amd.parallel import p2p
            p2p.ENABLED, p2p._default = True, ar
            torch.manual_seed(3)
            full = torch.randn(2 * world, 64, 17, 17, device="cuda").to(torch.bfloat16)
            bn = ops.BatchNorm2d(64, sync=True).cuda().train()
            bn_ref = ops.BatchNorm2d(64).cuda().train()
            x = full[2 * rank:2 * rank + 2].contiguous(memory_format=torch.channels_last).requires_grad_(True)
            xr = full.contiguous(memory_format=torch.channels_last).requires_grad_(True)
            y = bn(x, relu=True)
            yr = bn_ref(xr, relu=True)
            assert torch.allclose(y.float(), yr[2 * rank:2 * rank + 2].float(), atol=3e-2, rtol=3e-2)
            y.float().pow(2).sum().backward()
            yr.float().pow(2).sum().backward()
            assert torch.allclose(x.grad.float(), xr.grad[2 * rank:2 * rank + 2].float(), atol=5e-2, rtol=5e-2)
            assert torch.allclose(bn.running_mean, bn_ref.running_mean, atol=1e-3)
            # affine gradients: whatever each rank holds, the SUM over the ranks (world x DDP's average) is the
            # full-batch gradient
            for prm, ref in ((bn.weight, bn_ref.weight), (bn.bias, bn_ref.bias)):
                gsum = prm.grad.detach().cpu()
                dist.all_reduce(gsum)
                assert torch.allclose(gsum, ref.grad.cpu(), atol=2e-2 * float(ref.grad.abs().max()), rtol=2e-2)
            # and through a convolution's fused statistics (replicated partial sums all-reduced in place)
            conv = ops.Conv2d(64, 64, 3, padding=1, bias=False).cuda()
            torch.manual_seed(4)
            with torch.no_grad():
                conv.weight.copy_(torch.randn_like(conv.weight) * 0.05)
            bn2, bn2_ref = ops.BatchNorm2d(64, sync=True).cuda().train(), ops.BatchNorm2d(64).cuda().train()
            y2 = ops.conv_bn_act(conv, bn2, x.detach(), relu=True)
            y2r = ops.conv_bn_act(conv, bn2_ref, xr.detach(), relu=True)
            assert torch.allclose(y2.float(), y2r[2 * rank:2 * rank + 2].float(), atol=3e-2, rtol=3e-2)
        ar.check()
        if world == 2:
            # a peer that never arrives: ONE wall-clock bound, the error word raised at the next check, and the calls
            # that follow return at once (a step has ~200 of them) instead of waiting the bound out again each
            import time
            dist.barrier()
            if rank == 0:
                t = torch.ones(64, device="cuda")
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                ar.all_reduce_(t, timeout_ms=300)
                torch.cuda.synchronize()
                first = time.perf_counter() - t0
                with pytest.raises(RuntimeError, match="timed out waiting for a peer"):
                    ar.check()
                t0 = time.perf_counter()
                for _ in range(20):
                    ar.all_reduce_(t, timeout_ms=300)
                torch.cuda.synchronize()
                later = time.perf_counter() - t0
                assert 0.25 < first < 5.0 and later < 0.25, (first, later)
            dist.barrier()
    finally:
        ar.close()
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [1, 2, 4])
def test_one_shot_allreduce_across_processes(world):
    mp.spawn(_worker, args=(world, _free_port(), world == 2), nprocs=world, join=True)


def _engine_worker(rank, world, port, sync_bn):
    """Two ranks (processes) train DeepLabv3+ R18 through the full GPU path — arena sinks, aux-stream
    weight gradients, hook-launched buckets, SyncBN over the one-shot kernel — with every
    parallel.debug check on: bucket order, identical gradients after the all-reduce, identical
    replicas.  (Regression: gradient hooks counted twice emptied the bucket counters early.)"""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank), HSA_ENABLE_IPC_MODE_LEGACY="0", DPAMD_DIST_BACKEND="gloo",
                      DPAMD_P2P_TIMEOUT_MS="4000")
    # a rank that is still here after 90 s says where it is (every thread's stack on stderr) and exits: the full GPU
    # suite once stopped in this test for its whole 120 s limit and the parent's stack was all the log held
    import faulthandler
    faulthandler.dump_traceback_later(90, exit=True)
    from distributedpytorch_amd.models import build_model
    from distributedpytorch_amd.ops import SoftmaxCrossEntropy
    from distributedpytorch_amd.parallel import TrainEngine, debug, dist_env, init_distributed, p2p
    debug.ENABLED = True
    p2p.ENABLED, p2p.CHECK_EVERY = bool(sync_bn), 1
    info = init_distributed()
    try:
        torch.manual_seed(7 + rank)                      # different init per rank: the broadcast must fix it
        model = build_model("deeplabv3plus_resnet18", 21, sync_bn=sync_bn)
        eng = TrainEngine(model, SoftmaxCrossEntropy(255), lr=0.01, device=info.device, bucket_mb=4)
        assert len(eng.ddp.arena.buckets) >= 3
        g = torch.Generator().manual_seed(50 + rank)
        x = torch.randn(2, 3, 97, 97, generator=g).to(info.device)
        y = torch.randint(0, 21, (2, 97, 97), generator=g).to(info.device)
        losses = [float(eng.train_step(x, y)) for _ in range(4)]
        assert all(l == l for l in losses)
        assert eng.ddp.stats["buckets_launched"] == 4 * len(eng.ddp.arena.buckets)
        flat = eng.ddp.arena.flat_p.detach().cpu()
        other = flat.clone()
        dist.broadcast(other, src=0)
        assert torch.equal(flat, other), "replicas diverged"
        faulthandler.cancel_dump_traceback_later()
    finally:
        if p2p._default is not None:
            p2p._default.close()
            p2p._default = None
        dist_env.shutdown()


@pytest.mark.parametrize("sync_bn", [False, True])
def test_two_rank_engine_on_gpu_with_debug_checks(sync_bn):
    try:
        mp.spawn(_engine_worker, args=(2, _free_port(), sync_bn), nprocs=2, join=True)
    except Exception as e:
        # Two PROCESSES on one device: the one-shot kernels wait for each other's flags, i.e. need both processes'
        # kernels resident at the same moment — up to the queue scheduler here, a given on a node where every rank has
        # its own GPU.  A spin that was not served ends in the kernel's own wall-clock bound (handled: nothing
        # faults); that one outcome gets ONE more attempt, everything else fails at once.  (Never seen so far: the
        # stop this was first written for turned out to be gloo's device-tensor path in a debug check,
        # profiles/r3_two_rank_test_stop_once.md.)
        if not (sync_bn and "timed out waiting for a peer" in str(e)):
            raise
        mp.spawn(_engine_worker, args=(2, _free_port(), sync_bn), nprocs=2, join=True)


def _fold_sync_worker(rank, world, port):
    """The two algebraic folds under synchronised statistics (ops.sepconv: depthwise BN folded into the pointwise
    weights; ops.bnfold: a bottleneck's last BN folded into conv3's gradients): each rank runs its half of a batch,
    results must equal the unsynchronised modules on the whole batch."""
    import copy
    import faulthandler
    faulthandler.dump_traceback_later(75, exit=True)      # a stuck rank names its call site and ends the test
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      HSA_ENABLE_IPC_MODE_LEGACY="0")
    torch.cuda.set_device(rank % torch.cuda.device_count())
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from distributedpytorch_amd import ops
    from distributedpytorch_amd.models.resnet import Bottleneck
    from distributedpytorch_amd.models.xception import XceptionBlock
    from distributedpytorch_amd.ops import bnfold
    from distributedpytorch_amd.parallel import p2p
    from distributedpytorch_amd.parallel.p2p import OneShotAllReduce
    ar = OneShotAllReduce(None)
    p2p.ENABLED, p2p._default = True, ar
    cl = torch.channels_last
    try:
        for case in ("xception", "bottleneck"):
            torch.manual_seed(21)
            if case == "xception":
                net = torch.nn.Sequential(XceptionBlock(64, 64, 2, dilation=2), XceptionBlock(64, 64, 2, dilation=1))
            else:
                b0, b1 = Bottleneck(64, 16), Bottleneck(64, 16)
                b0.fold_tail, b1.input_exclusive = True, True
                net = torch.nn.Sequential(ops.Conv2d(64, 64, 1, bias=False), b0, b1)
            with torch.no_grad():
                for m in net.modules():
                    if hasattr(m, "running_mean"):
                        m.weight.uniform_(0.6, 1.4); m.bias.normal_(0, 0.5)
            ref = copy.deepcopy(net).cuda().train()
            net = ops.convert_sync_batchnorm(net).cuda().train()
            full = torch.randn(2 * world, 64, 19, 17, device="cuda").to(torch.bfloat16)
            gfull = torch.randn(2 * world, 64, 19, 17, device="cuda").to(torch.bfloat16)
            sl = slice(2 * rank, 2 * rank + 2)
            x = full[sl].contiguous(memory_format=cl).requires_grad_(True)
            xr = full.contiguous(memory_format=cl).requires_grad_(True)
            bnfold.COUNTS.update(folded=0, fallback=0)
            y = net(x)
            y.backward(gfull[sl].contiguous(memory_format=cl))
            if case == "bottleneck":
                assert bnfold.COUNTS["folded"] == 1, bnfold.COUNTS
            yr = ref(xr)
            yr.backward(gfull.contiguous(memory_format=cl))
            torch.cuda.synchronize()

            def cos(a, b):
                a, b = a.double().flatten(), b.double().flatten()
                return float((a @ b) / (a.norm() * b.norm()).clamp_min(1e-30))
            assert cos(y.float(), yr[sl].float()) > 0.9999, (case, cos(y.float(), yr[sl].float()))
            assert cos(x.grad.float(), xr.grad[sl].float()) > 0.995, (case, cos(x.grad.float(), xr.grad[sl].float()))
            for (k, prm), (_, rp) in zip(net.named_parameters(), ref.named_parameters()):
                # (on the host: gloo's device-tensor path stalled now and then with two processes on one GPU — both
                # ranks inside this all_reduce, no kernel outstanding; the watchdog above named the line)
                gsum = prm.grad.detach().float().cpu()
                dist.all_reduce(gsum)                       # world x DDP's average = the full-batch gradient
                if float(rp.grad.norm()) > 1e-3:
                    assert cos(gsum, rp.grad.float().cpu()) > 0.995, (case, k, cos(gsum, rp.grad.float().cpu()))
            for (k, b), (_, rb) in zip(net.named_buffers(), ref.named_buffers()):
                if k.endswith("running_mean") or k.endswith("running_var"):
                    assert torch.allclose(b, rb, atol=2e-2, rtol=2e-2), (case, k)
        ar.check()
    finally:
        faulthandler.cancel_dump_traceback_later()
        ar.close()
        p2p._default = None
        dist.destroy_process_group()


def test_folded_paths_under_syncbn_equal_global_batch():
    mp.spawn(_fold_sync_worker, args=(2, _free_port()), nprocs=2, join=True)


def test_entrypoint_two_ranks_master_alone_in_a_forward(tmp_path):
    """`train_pascal.py --gpus 2` (two ranks sharing the device, gloo) through training, validation with per-rank tail
    batches, the MASTER-ONLY validation figure and the checkpoint barrier.  The figure's forward meets problems no
    table holds on rank 0 alone: a tile choice that cost a collective there (round 3's rank-0-decides broadcast, before
    it was confined to `collective_tuning` sections) left rank 0 in a broadcast and rank 1 in the barrier."""
    import json
    import subprocess
    env = dict(os.environ, DPAMD_DIST_BACKEND="gloo", MASTER_ADDR="127.0.0.1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, "-u", os.path.join(ROOT, "train_pascal.py"), "--synthetic", "--model",
                        "deeplabv3plus_resnet18", "--crop", "129", "--trainBatch", "8", "--testBatch", "8",
                        "--synthetic-len", "72", "--nEpochs", "2", "--workers", "0", "--log-every", "0",
                        "--hang-timeout", "90", "--gpus", "2", "--save-dir", str(tmp_path)],
                       capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-2500:])
    recs = [json.loads(l) for l in (tmp_path / "run_0" / "log.jsonl").read_text().splitlines()]
    vals = [x for x in recs if x.get("kind") == "val"]
    assert len(vals) == 2 and all(v["path"] == "device" for v in vals)
    assert (tmp_path / "run_0" / "models").exists()


def _graph_step_ddp_worker(rank, world, port):
    """Two ranks, graph-step: forward + backward replayed from a hipGraph, bucket collectives and the SGD update
    launched by the host after each replay.  Deterministic mode, dropout off: bit-identical to the eager two-rank
    engine on the same batches, replicas identical."""
    import copy
    import faulthandler
    faulthandler.dump_traceback_later(100, exit=True)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank), HSA_ENABLE_IPC_MODE_LEGACY="0", DPAMD_DIST_BACKEND="gloo")
    from distributedpytorch_amd.models import build_model
    from distributedpytorch_amd.ops import SoftmaxCrossEntropy, _backend
    from distributedpytorch_amd.parallel import TrainEngine, dist_env, init_distributed
    info = init_distributed()
    try:
        torch.manual_seed(3)
        base = build_model("deeplabv3plus_resnet18", 21)
        for m in base.modules():
            if isinstance(getattr(m, "p_drop", None), tuple):
                m.p_drop = (0.0, 0.0)
            elif hasattr(m, "p_drop"):
                m.p_drop = 0.0
        g = torch.Generator().manual_seed(50 + rank)
        xs = [torch.randn(2, 3, 97, 97, generator=g).to(info.device) for _ in range(7)]
        ys = [torch.randint(0, 21, (2, 97, 97), generator=g).to(info.device) for _ in range(7)]
        res = {}
        with _backend.deterministic_mode(True):
            for mode in ("eager", "graph"):
                eng = TrainEngine(copy.deepcopy(base), SoftmaxCrossEntropy(255), lr=0.01, momentum=0.9, device=info.device,
                                  bucket_mb=4, graph_step=mode == "graph")
                losses = []
                for i, (x, y) in enumerate(zip(xs, ys)):
                    eng.optimizer.param_groups[0]["lr"] = 0.01 * (1.0 - i / 10.0)
                    losses.append(float(eng.train_step(x, y)))
                torch.cuda.synchronize()
                res[mode] = (losses, eng.ddp.arena.flat_p.detach().cpu().clone(), len(eng._graphs),
                             eng.ddp.stats["buckets_launched"], len(eng.ddp.arena.buckets))
        (l1, p1, n1, b1, nb), (l2, p2, n2, b2, _) = res["eager"], res["graph"]
        assert n1 == 0 and n2 == 1 and b1 == b2 == 7 * nb, (n1, n2, b1, b2, nb)
        assert l1 == l2, (rank, l1, l2)
        assert torch.equal(p1, p2), "graph-step replicas differ from the eager ones"
        other = p2.clone()
        dist.broadcast(other, src=0)
        assert torch.equal(p2, other), "replicas diverged"
    finally:
        dist_env.shutdown()


def test_graph_step_with_two_ranks_equals_the_eager_engine():
    mp.spawn(_graph_step_ddp_worker, args=(2, _free_port()), nprocs=2, join=True)
