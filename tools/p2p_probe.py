#!/usr/bin/env python
"""Probe / micro-benchmark of the one-shot peer-to-peer all-reduce (parallel.p2p).  Launch with torchrun;
on a 1-GPU box the ranks share device 0 and talk over gloo for the handshake (IPC mapping between
processes still goes through hipIpc*, the kernels still exchange granules through each other's buffers)."""
import faulthandler
import os
import sys
import time

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
faulthandler.dump_traceback_later(90, exit=True)
rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
ngpu = torch.cuda.device_count()
torch.cuda.set_device(rank % ngpu)
backend = os.environ.get("DPAMD_DIST_BACKEND", "nccl" if ngpu >= world else "gloo")
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29544")
dist.init_process_group(backend, rank=rank, world_size=world)
from distributedpytorch_amd.parallel.p2p import OneShotAllReduce  # noqa: E402

T0 = time.perf_counter()
ar = OneShotAllReduce(None)
print(f"[rank {rank}] handshake done ({time.perf_counter() - T0:.1f} s)", flush=True)
ok = True
for it, n in enumerate([1, 7, 64, 513, 2048, 4097, 8192] * 6):
    base = torch.arange(n, device="cuda", dtype=torch.float32) * 0.25 + it
    t = base * (rank + 1)
    ar.all_reduce_(t)
    want = base * sum(r + 1 for r in range(world))
    if not torch.equal(t, want):
        ok = False
        print(f"[rank {rank}] MISMATCH n={n} it={it}: max err {(t - want).abs().max().item()}", flush=True)
ar.check()
print(f"[rank {rank}] correctness loop done: {'ok' if ok else 'WRONG'} ({time.perf_counter() - T0:.1f} s)", flush=True)
# bit-identical across ranks with non-trivial values
g = torch.Generator(device="cuda").manual_seed(5 + rank)
t = torch.randn(4096, device="cuda", generator=g)
ar.all_reduce_(t)
ref = t.clone()
dist.broadcast(ref, src=0)
same = torch.equal(ref, t)
ar.check()
# latency
x = torch.ones(4096, device="cuda")
for _ in range(20):
    ar.all_reduce_(x)
torch.cuda.synchronize()
dist.barrier()
t0 = time.perf_counter()
iters = int(os.environ.get("P2P_ITERS", "300"))
print(f"[rank {rank}] bit-identical {same} ({time.perf_counter() - T0:.1f} s)", flush=True)
for _ in range(iters):
    ar.all_reduce_(x)
torch.cuda.synchronize()
p2p_us = (time.perf_counter() - t0) / iters * 1e6
print(f"[rank {rank}] one-shot p2p {p2p_us:.1f} us/call ({time.perf_counter() - T0:.1f} s)", flush=True)
ref_us = float("nan")
if os.environ.get("P2P_SKIP_REF", "0") == "0":
    y = torch.ones(4096, device="cuda")
    for _ in range(5):
        dist.all_reduce(y)
    torch.cuda.synchronize()
    dist.barrier()
    t0 = time.perf_counter()
    for _ in range(100):
        dist.all_reduce(y)
    torch.cuda.synchronize()
    ref_us = (time.perf_counter() - t0) / 100 * 1e6
ar.check()
print(f"[rank {rank}/{world}] values {'ok' if ok else 'WRONG'}, bit-identical across ranks: {same}; "
      f"16 KB all-reduce: one-shot p2p {p2p_us:.1f} us/call, {backend} {ref_us:.1f} us/call", flush=True)
ar.close()
dist.destroy_process_group()
sys.exit(0 if ok and same else 1)
