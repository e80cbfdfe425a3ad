"""Training / evaluation engine: the reference's module-level loop
(`/root/reference/train_pascal.py:180-308`) and its half-written ``train_epoch`` refactor
(`:311-354`, TODO "# single fn train / # single fn val" `:1-2`) as a reusable object.

One engine = one rank.  ``train_step`` is the hot path of SURVEY.md §3.1 re-designed:

    batch already on device (GPU augmentation, no per-step host asserts, no ``.item()``)
    forward  (HIP kernels, bf16 NHWC)
    loss     (fused ignore-index softmax-CE)
    backward (bucket all-reduces launch from hooks and overlap the remaining kernels)
    finish   (wait buckets) → fused SGD (1/world scale, bf16 refresh, grad zero)

The loss is returned as a device tensor; callers decide when to pay for a host read.
"""
from __future__ import annotations

import os

from typing import Callable, Dict, Iterable, Optional, Sequence

import torch
import torch.nn as nn

from ..ops import confusion_matrix, miou_from_confusion, workspace
from ..utils.timers import PhaseTimer
from . import dist_env
from .ddp import DistributedDataParallel
from .optim import FlatSGD


def named_lr_groups(model: nn.Module, head_lr_mult: float = 1.0):
    """Split parameters into the reference's 1×/10× groups (`train_pascal.py:90-91`)."""
    if head_lr_mult == 1.0 or not hasattr(model, "get_10x_lr_params"):
        return [list(model.named_parameters())], [1.0]
    head_ids = {id(p) for p in model.get_10x_lr_params()}
    base = [(n, p) for n, p in model.named_parameters() if id(p) not in head_ids]
    head = [(n, p) for n, p in model.named_parameters() if id(p) in head_ids]
    return [base, head], [1.0, head_lr_mult]


class _NullTimerT:
    from contextlib import contextmanager as _cm

    @_cm
    def phase(self, name: str):
        yield


_NullTimer = _NullTimerT()


class TrainEngine:
    def __init__(self, model: nn.Module, criterion: Callable, lr: float = 0.007, momentum: float = 0.9,
                 weight_decay: float = 5e-4, head_lr_mult: float = 1.0,
                 device: Optional[torch.device] = None, compute_dtype: Optional[torch.dtype] = None,
                 bucket_mb: int = 64, accum_steps: int = 1, process_group=None, markers: bool = False,
                 graph_step: bool = False):
        self.device = device or torch.device("cpu")
        self.on_gpu = self.device.type == "cuda"
        self.compute_dtype = compute_dtype or (torch.bfloat16 if self.on_gpu else torch.float32)
        model = model.to(self.device)
        if self.on_gpu:
            model = model.to(memory_format=torch.channels_last)
        groups, mults = named_lr_groups(model, head_lr_mult)
        lowp = self.compute_dtype if self.compute_dtype != torch.float32 else None
        self.ddp = DistributedDataParallel(model, groups, bucket_bytes=bucket_mb << 20, lowp_dtype=lowp,
                                           process_group=process_group)
        self.model = model
        self.criterion = criterion
        if getattr(criterion, "fuses_upsample", False) and hasattr(model, "defer_upsample"):
            model.defer_upsample = True
        adapters = [(name + "." if name else "", m) for name, m in model.named_modules()
                    if hasattr(m, "state_to_logical") and hasattr(m, "state_to_physical")]
        self.optimizer = FlatSGD(self.ddp.arena, [lr * m for m in mults], momentum, weight_decay, adapters=adapters)
        self.accum_steps = max(int(accum_steps), 1)
        self._micro = 0
        self.timer = PhaseTimer(self.on_gpu, markers)
        self.global_step = 0
        self._step_events: list = []      # bounds how far the host may run ahead of the GPU
        # graph_step: the whole step (forward, loss, backward, fused SGD, transposing refresh) replayed from ONE hipGraph
        # per input shape (see _train_step_graphed).  Below ~B32 the R101 step is bound by the host's ~900 launches
        # (B8 and B16 both take 29 ms eagerly); a replay costs what the device needs.
        self.graph_step = bool(graph_step) and self.on_gpu
        self._graphs: dict = {}
        self._graph_warm: dict = {}
        if self.graph_step:
            if self.accum_steps != 1:
                raise NotImplementedError("graph_step: no gradient accumulation (no_sync micro-steps are decided on the host)")
            if self.ddp._comm and any(getattr(m, "sync", False) for m in model.modules()):
                raise NotImplementedError("graph_step with several ranks: no synchronised BatchNorm (its collectives "
                                          "would sit inside the captured forward and backward)")
            # Several ranks: the capture holds forward + backward only; the bucket collectives and the SGD update are
            # launched by the host after every replay (a handful of launches).  RCCL collectives INSIDE a capture were
            # tried on a forced 1-rank group: the process group's watchdog thread queries the events of the captured
            # collectives (hipErrorCapturedEvent, abort) — and gloo's run on the host anyway.  The price is that the
            # gradient exchange no longer overlaps backward: ~236 MB for R101, about a millisecond over xGMI.
        # Stream priorities (history; ops/streams.py has the current state): a LOW-priority weight-gradient stream
        # was worth 951 -> 957-958 img/s at B128 in round 1 and nothing measurable on R101 at the end of round 3, and
        # stalled a step in a process with several other streams — every stream runs at normal priority now.  A
        # HIGH-priority step stream (961-962) was never built in: RCCL's kernels run at normal priority and would be
        # dispatched behind every compute kernel, and the one-shot SyncBN kernel (csrc/p2p_allreduce.hip) SPINS until
        # every peer's granules arrive — on a queue that outranks the producer of those granules (ranks sharing a
        # device) nothing progresses until the bounded spin times out.
    # ------------------------------------------------------------------
    def prepare_input(self, images: torch.Tensor) -> torch.Tensor:
        x = images.to(self.device, non_blocking=True)
        if x.dtype != self.compute_dtype:
            x = x.to(self.compute_dtype)
        if self.on_gpu and x.dim() == 4:
            x = x.contiguous(memory_format=torch.channels_last)
        return x

    def train_step(self, images: torch.Tensor, target: torch.Tensor) -> torch.Tensor:
        self.model.train()
        x = self.prepare_input(images)
        # TODO `train_pascal.py:5`: "label non_blocking = True"
        y = target.to(self.device, non_blocking=True)
        if self.graph_step:
            return self._train_step_graphed(x, y)
        return self._train_step_eager(x, y)

    def _bound_run_ahead(self) -> None:
        # The loss is never read on the host, so nothing stops Python from queueing steps faster than
        # the GPU retires them, and the caching allocator then grows without bound (B128: 280 GiB, OOM).
        # Wait for the step before the previous one: two steps in flight keep the queues full.
        if self.on_gpu and len(self._step_events) >= 2:
            self._step_events.pop(0).synchronize()

    def _mark_step_end(self) -> None:
        if self.on_gpu:
            ev = torch.cuda.Event()
            ev.record()
            self._step_events.append(ev)

    def _train_step_eager(self, x: torch.Tensor, y: torch.Tensor, capturing=False) -> torch.Tensor:
        """``capturing``: False (an eager step), "whole" (recorded into a graph: forward ... SGD) or "fwd_bwd" (several
        ranks: forward + backward recorded, gradient hooks silent; _finish_replayed_step does the rest on the host)."""
        last_micro = (self._micro + 1) % self.accum_steps == 0
        if not capturing:
            self._bound_run_ahead()
        timer = _NullTimer if capturing else self.timer           # timed events do not belong into a capture
        workspace.begin_step(self.device)   # one memset re-arms every reduction accumulator of the step
        # forward + backward of a training step: the one place where every rank runs the same launches in the same
        # order, hence the only place where a new problem's tile choice may be a collective (rank 0 measures)
        from ..ops import _hip_conv
        with _hip_conv.collective_tuning():
            with timer.phase("fwd"):
                out = self.ddp(x)
                loss = self.criterion(out, y)
                if self.accum_steps > 1:
                    loss = loss / self.accum_steps
            with timer.phase("bwd"):
                if last_micro and capturing != "fwd_bwd":
                    loss.backward()
                else:
                    with self.ddp.no_sync():
                        loss.backward()
        if capturing == "fwd_bwd":
            from ..ops import streams
            streams.join(self.device)            # the weight-gradient stream rejoins the capturing stream
            return loss.detach()
        self._micro += 1
        if last_micro:
            with timer.phase("comm_wait"):
                scale = self.ddp.finish_gradient_sync()
            from . import debug
            if debug.ENABLED and not capturing:
                debug.check_streams_joined(self.device)
            with timer.phase("optim"):
                self.optimizer.step(grad_scale=scale, zero_grad=True)
            self.global_step += 1
            if not capturing:
                self._mark_step_end()
            if debug.ENABLED and not capturing and self.global_step % 10 == 1:
                # running statistics are rank-local unless every BatchNorm is synchronised
                bns = [m for m in self.model.modules() if hasattr(m, "running_mean") and hasattr(m, "sync")]
                all_sync = bool(bns) and all(m.sync for m in bns)
                debug.check_replicas_equal(self.ddp.arena.flat_p, self.model.buffers() if all_sync else ())
            from . import p2p
            if p2p._default is not None and not capturing:
                if self.global_step == 1:
                    p2p.check_first_step()    # a timed-out (partial) reduction must never feed a second step
                elif self.global_step % p2p.CHECK_EVERY == 0:
                    p2p._default.check()      # a peer that never arrived (bounded spin) surfaces here
        return loss.detach()

    # ------------------------------------------------------------------ the step as one hipGraph
    _GRAPH_WARMUP = 3

    def _train_step_graphed(self, x: torch.Tensor, y: torch.Tensor) -> torch.Tensor:
        """One hipGraph per input shape.  The first steps of a shape run eagerly (tile choices, lazy initialisation, the
        first-step form of the momentum update), the next one is captured — a capture records, it does not execute — and
        replayed at once, so every batch trains exactly once.  What changes from step to step cannot be a kernel
        argument (frozen at capture): the inputs are copied into static buffers, the learning rate reaches the SGD
        kernels through one device scalar per parameter group that the schedule writes, and the fused dropouts mix a
        device counter, incremented by the captured step itself, into their keys (csrc/common.h: g_dp_drop_word)."""
        key = (tuple(x.shape), x.dtype, tuple(y.shape), y.dtype)
        st = self._graphs.get(key)
        if st is None:
            n = self._graph_warm.get(key, 0)
            if n < self._GRAPH_WARMUP:
                self._graph_warm[key] = n + 1
                return self._train_step_eager(x, y)
            st = self._capture_step(x, y)
            self._graphs[key] = st
        self._bound_run_ahead()
        if st["x"].data_ptr() != x.data_ptr():
            st["x"].copy_(x)
        if st["y"].data_ptr() != y.data_ptr():
            st["y"].copy_(y)
        if st["mode"] == "whole":
            for g, pg in enumerate(self.optimizer.param_groups):      # (fill kernels: no host-to-device copy)
                st["lr_dev"][g:g + 1].fill_(float(pg["lr"]))
            st["graph"].replay()
            self.optimizer._steps += 1
        else:
            st["graph"].replay()
            # several ranks: every bucket is launched now (no hook fired during the replay), then the update
            scale = self.ddp.finish_gradient_sync()
            self.optimizer.step(grad_scale=scale, zero_grad=True)
        self.global_step += 1
        self._mark_step_end()
        return st["loss"]

    def _capture_step(self, x: torch.Tensor, y: torch.Tensor) -> dict:
        from ..ops import _backend
        lib = _backend.lib()
        st = {"x": x.clone(memory_format=torch.preserve_format), "y": y.clone(),
              "lr_dev": torch.zeros(len(self.optimizer.param_groups), dtype=torch.float32, device=self.device),
              "drop_word": torch.zeros(1, dtype=torch.int32, device=self.device)}
        st["mode"] = "fwd_bwd" if self.ddp._comm else "whole"
        steps_before, global_before, micro_before = self.optimizer._steps, self.global_step, self._micro
        torch.cuda.synchronize(self.device)
        graph = torch.cuda.CUDAGraph()
        lib.dp_set_drop_word(st["drop_word"].data_ptr())
        self.optimizer.lr_dev = st["lr_dev"] if st["mode"] == "whole" else None
        try:
            # thread_local: calls that are unsafe during a capture are checked for THIS thread only — a loader's
            # pin-memory thread (host data path, stream mode) allocating or copying next to the capture invalidated it
            # in the default (global) mode; the autograd thread's launches go to the capturing stream either way
            with torch.cuda.graph(graph, capture_error_mode="thread_local"):
                st["drop_word"].add_(1)
                st["loss"] = self._train_step_eager(st["x"], st["y"], capturing=st["mode"])
        finally:
            lib.dp_set_drop_word(None)            # the pointers are kernel arguments of the capture now; eager
            self.optimizer.lr_dev = None          # launches (validation, another engine) see none
        # nothing ran: undo the bookkeeping of the recorded step (the replay that follows does its own)
        self.optimizer._steps, self.global_step, self._micro = steps_before, global_before, micro_before
        st["graph"] = graph
        return st

    def inference_forward(self, graph: bool = False):
        """The model's forward for evaluation loops (call ``model.eval()`` first): the module itself, or — ``graph`` on a
        GPU — its replay from a hipGraph per input shape (parallel.graphs; the result is valid until the next call)."""
        if graph and self.on_gpu:
            if getattr(self, "_graphed", None) is None:
                from .graphs import GraphedForward
                self._graphed = GraphedForward(self.model)
            return self._graphed
        return self.model

    @torch.no_grad()
    def evaluate(self, loader: Iterable, num_classes: int, ignore_index: int = 255,
                 max_batches: Optional[int] = None, graph: bool = False) -> Dict[str, float]:
        """21-class mIoU from a device confusion matrix, merged across ranks (K12 + C6).

        ``graph=True`` (GPU): the forward is replayed from a hipGraph per input shape — the
        reference's batch-1 evaluation is launch-bound (parallel.graphs)."""
        self.model.eval()
        fwd = self.inference_forward(graph)
        cm = torch.zeros(num_classes, num_classes, dtype=torch.int64, device=self.device)
        loss_sum = torch.zeros((), dtype=torch.float64, device=self.device)
        n = 0
        for i, batch in enumerate(loader):
            if max_batches is not None and i >= max_batches:
                break
            images, target = batch if isinstance(batch, (tuple, list)) else (batch["image"], batch["label"])
            x = self.prepare_input(images)
            y = target.to(self.device, non_blocking=True)
            out = fwd(x)
            loss_sum += self.criterion(out, y).double()
            logits = out[0] if isinstance(out, (tuple, list)) else out
            confusion_matrix(logits, y, num_classes, ignore_index, out=cm)
            n += 1
        if dist_env.get_world_size() > 1:
            import torch.distributed as dist
            packed = torch.cat([cm.view(-1).double(), loss_sum.view(1), loss_sum.new_tensor([float(n)])])
            dist_env.all_reduce_(packed)         # (host-staged under gloo)
            cm = packed[:-2].view(num_classes, num_classes).round().long()
            loss_sum, n = packed[-2], float(packed[-1].item())
        miou, iou, acc = miou_from_confusion(cm)
        return {"miou": miou, "pixel_acc": acc, "loss": float(loss_sum.item()) / max(n, 1),
                "per_class_iou": [float(v) for v in iou.tolist()]}
