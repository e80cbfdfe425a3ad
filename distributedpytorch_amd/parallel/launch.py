"""Self-launch: one command starts one process per GPU.

The reference drives its 4 GPUs from a single ``python train_pascal.py`` (one process, threads of
``nn.DataParallel``, `/root/reference/train_pascal.py:92`); its TODO header (`:7-8`) names the missing
pieces of the per-process design — device from the local rank, an explicit rendezvous port.  Here the
same one command works: ``python bench.py --gpus 8`` / ``python train_pascal.py --gpus 8``.  The
PARENT never touches the GPU: it counts the devices from the KFD topology in sysfs (no HIP call at all —
``torch.cuda.device_count()`` falls through to ``hipGetDeviceCount`` when ``amdsmi`` is absent, which starts
the HIP runtime — so that is only the fall-back when sysfs has no topology), picks a free rendezvous port,
starts N fresh children with the torchrun environment contract (``RANK``, ``LOCAL_RANK``,
``WORLD_SIZE``, ``MASTER_ADDR``, ``MASTER_PORT``), waits for them, and exits with the first non-zero
status after stopping the survivors — a dead rank must not leave the others blocked in a collective.
When the environment already carries ``WORLD_SIZE`` (torchrun, or one of our own children) nothing
is launched and the caller continues as a rank.

Each child binds to ``cuda:LOCAL_RANK`` (`parallel.dist_env.init_distributed`).  ``ROCR``/``HIP``
visibility masks are left alone so RCCL sees the whole xGMI mesh and can map peer buffers.
"""
from __future__ import annotations

import os
import signal
import socket
import subprocess
import sys
import time
from typing import List, Optional, Sequence

# the platform only supports dmabuf IPC; RCCL and HIP IPC handles need this BEFORE the runtime starts
IPC_ENV = {"HSA_ENABLE_IPC_MODE_LEGACY": "0"}


def free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def already_a_rank() -> bool:
    return "WORLD_SIZE" in os.environ and "RANK" in os.environ


KFD_NODES = "/sys/class/kfd/kfd/topology/nodes"


def _mask_count(total: int) -> int:
    """Apply the visibility masks a child would see (comma lists of indices; UUID entries count as one each)."""
    for var in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v is None:
            continue
        items = [t for t in v.split(",") if t.strip() != ""]
        n = 0
        for t in items:
            t = t.strip()
            if t.lstrip("-").isdigit() and not 0 <= int(t) < total:
                break                                    # the runtime stops at the first invalid index
            n += 1
        total = min(total, n)
    return total


def kfd_gpu_count(root: str = KFD_NODES) -> Optional[int]:
    """GPU nodes of the KFD topology: a node is a GPU when its ``properties`` file has ``simd_count`` > 0 (CPU
    nodes carry 0).  None when the topology cannot be read (no amdgpu driver, restricted container)."""
    try:
        nodes = sorted(os.listdir(root))
    except OSError:
        return None
    n, readable = 0, False
    for node in nodes:
        try:
            with open(os.path.join(root, node, "properties")) as f:
                props = dict(line.split()[:2] for line in f if len(line.split()) >= 2)
        except OSError:
            continue
        readable = True
        if int(props.get("simd_count", "0")) > 0:
            n += 1
    return n if readable else None


def visible_gpus() -> int:
    """Number of devices a child will see, without starting the HIP runtime in this process when sysfs can tell."""
    n = kfd_gpu_count()
    if n is not None:
        # a container may show the host's whole topology but pass only some render nodes through
        import glob
        nodes = glob.glob("/dev/dri/renderD*")
        if nodes:
            n = min(n, sum(os.access(d, os.R_OK | os.W_OK) for d in nodes))
        return _mask_count(n)
    try:
        import torch
        return int(torch.cuda.device_count())            # starts the runtime; the parent never execs, so harmless
    except Exception:
        return 0


def child_env(rank: int, world: int, port: int, base: Optional[dict] = None) -> dict:
    env = dict(os.environ if base is None else base)
    env.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), LOCAL_WORLD_SIZE=str(world),
               MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), DPAMD_SELF_LAUNCHED="1")
    for k, v in IPC_ENV.items():
        env.setdefault(k, v)
    # leave a few host threads to every rank instead of world x all cores
    if "OMP_NUM_THREADS" not in env:
        env["OMP_NUM_THREADS"] = str(max(1, (os.cpu_count() or 8) // max(world, 1)))
    return env


def _stop(procs: List[subprocess.Popen], grace_s: float = 5.0) -> None:
    """Terminate exactly the children we started (never by pattern)."""
    for p in procs:
        if p.poll() is None:
            try:
                p.send_signal(signal.SIGTERM)
            except OSError:
                pass
    t_end = time.monotonic() + grace_s
    for p in procs:
        while p.poll() is None and time.monotonic() < t_end:
            time.sleep(0.05)
        if p.poll() is None:
            try:
                p.kill()
            except OSError:
                pass
            p.wait()


def launch_ranks(n: int, cmd: Sequence[str], timeout_s: Optional[float] = None,
                 env: Optional[dict] = None) -> int:
    """Start ``n`` copies of ``cmd`` as ranks 0..n-1 and return the job's exit status (0 only when
    every rank exited 0).  Rank 0 inherits stdout (it prints the result line); every rank inherits
    stderr."""
    port = free_port()
    procs: List[subprocess.Popen] = []
    t0 = time.monotonic()
    try:
        for r in range(n):
            procs.append(subprocess.Popen(list(cmd), env=child_env(r, n, port, env),
                                          stdout=None if r == 0 else sys.stderr))
        status = 0
        while True:
            alive = 0
            for r, p in enumerate(procs):
                rc = p.poll()
                if rc is None:
                    alive += 1
                elif rc != 0 and status == 0:
                    status = rc if rc > 0 else 128 - rc
                    print(f"[launch] rank {r} exited with status {rc}; stopping the other ranks",
                          file=sys.stderr, flush=True)
            if status != 0 or alive == 0:
                break
            if timeout_s is not None and time.monotonic() - t0 > timeout_s:
                print(f"[launch] {n}-rank job exceeded {timeout_s:.0f} s; stopping it", file=sys.stderr, flush=True)
                status = 124
                break
            time.sleep(0.1)
        return status
    finally:
        _stop(procs)


def maybe_self_launch(n_gpus: int, argv: Optional[Sequence[str]] = None, script: Optional[str] = None,
                      cpu: bool = False, timeout_s: Optional[float] = None) -> Optional[int]:
    """Call FIRST in ``main`` (before anything that initialises the GPU).

    Returns ``None`` when the caller should simply continue (it is a rank already, or one process is
    all that was asked for); otherwise launches ``n_gpus`` ranks of the same script + arguments and
    returns the job's exit status, which the caller passes to ``sys.exit``."""
    if already_a_rank() or n_gpus <= 1:
        return None
    if not cpu and os.environ.get("DPAMD_DIST_BACKEND", "") != "gloo":
        have = visible_gpus()
        if have < n_gpus:
            print(f"[launch] --gpus {n_gpus} but only {have} GPU(s) are visible "
                  f"(set DPAMD_DIST_BACKEND=gloo to rehearse with ranks sharing devices)", file=sys.stderr)
            return 2
    script = script or os.path.abspath(sys.argv[0])
    args = list(sys.argv[1:] if argv is None else argv)
    return launch_ranks(n_gpus, [sys.executable, "-u", script, *args], timeout_s=timeout_s)
