#!/usr/bin/env python
"""Probe: do gloo collectives on GPU tensors work when several ranks share ONE device?
(Decides whether a multi-rank rehearsal of the GPU data-parallel path over gloo means anything on a
1-GPU box.)  Launch with torchrun, 2 ranks."""
import faulthandler
import os
import sys

import torch
import torch.distributed as dist

faulthandler.dump_traceback_later(40, exit=True)
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
torch.cuda.set_device(0)
dist.init_process_group("gloo", rank=rank, world_size=world)
for numel in (16, 1 << 20, 16 << 20):
    t = torch.full((numel,), float(rank + 1), device="cuda")
    dist.all_reduce(t)
    torch.cuda.synchronize()
    print(f"[rank {rank}] sync all_reduce {numel}: {t[0].item()}", flush=True)
    t = torch.full((numel,), float(rank + 1), device="cuda")
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        u = t * 2
    torch.cuda.current_stream().wait_stream(s)
    w = dist.all_reduce(u, async_op=True)
    w.wait()
    torch.cuda.synchronize()
    print(f"[rank {rank}] async all_reduce {numel}: {u[0].item()}", flush=True)
dist.destroy_process_group()
print(f"[rank {rank}] done", flush=True)
