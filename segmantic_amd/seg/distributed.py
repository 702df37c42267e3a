"""Data-parallel plumbing: one process per GPU, RCCL gradient all-reduce over xGMI.

The reference gets DDP implicitly from ``pl.Trainer(devices=len(gpu_ids))``
(``src/segmantic/seg/monai_unet.py:529-538``): one process per GPU, gradient all-reduce (mean)
inside ``loss.backward()``, BatchNorm statistics NOT synchronised.  Here the engine's explicit
backward fills one flat f32 gradient arena in exactly reverse parameter order, so completed
suffixes of the arena are all-reduced in buckets on a side HIP stream while the rest of the
backward still runs; the fused optimiser applies the 1/world mean through its grad scale.
"""
from __future__ import annotations

import os
from typing import List, Optional

import torch
import torch.distributed as dist

from . import streams


def env_world() -> tuple:
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
            int(os.environ.get("WORLD_SIZE", "1")))


def rank_device_index(gpu_ids, local_rank: int) -> int:
    """GPU of a rank: entry ``local_rank`` of ``gpu_ids`` (wrapping), or ``local_rank`` itself."""
    ids = [int(g) for g in (gpu_ids or [])]
    if ids:
        return ids[local_rank % len(ids)]
    return local_rank % max(1, torch.cuda.device_count())


def init_distributed(backend: Optional[str] = None, device_index: Optional[int] = None) -> tuple:
    """Initialise torch.distributed from the torchrun environment (idempotent).  ``device_index``:
    the GPU this rank computes on (default ``LOCAL_RANK``); RCCL binds its communicator to it."""
    rank, local_rank, world = env_world()
    if world > 1 and not dist.is_initialized():
        # (GPU_MAX_HW_QUEUES -- a rank drives more streams than ROCm's default 4 hardware queues -- is
        # set when the package is imported, segmantic_amd/__init__.py, i.e. before any device call of
        # any entry point; setting it here was too late for train())
        if backend is None:
            # SEGMI_DIST_BACKEND=gloo: rehearse the multi-process path where RCCL cannot run (several
            # ranks sharing one GPU, CPU-only boxes); the product default on GPUs is nccl = RCCL
            backend = os.environ.get("SEGMI_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend == "nccl":
            torch.cuda.set_device(rank_device_index(None, local_rank) if device_index is None
                                  else int(device_index))
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, local_rank, world


class GradSync:
    """Bucketed all-reduce (SUM) of a flat gradient arena, overlapped with backward.

    ``ready(lo)`` is called by the engine when every gradient at arena offset >= lo is final.
    Buckets are cut from the end of the arena (backward order); each is reduced on a side
    stream as soon as it is complete.  ``finish()`` makes the compute stream wait for all of
    them.  Bucket size: the 19 MB arena of the benchmark network goes out as 4 buckets of 4 MiB + the
    tail (1.3 / 0.85 / 0.2 MiB as the down path finishes, 30 KiB at finish()); on xGMI (point-to-point
    links) fewer, larger messages are per-link-bound rather than latency-bound.
    """

    def __init__(self, flat_grad: torch.Tensor, bucket_bytes: int = 4 << 20, group=None):
        self.flat_grad = flat_grad
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.group = group
        self.bucket_elems = max(1, bucket_bytes // 4)
        self.tail_min_elems = min(self.bucket_elems, 16384)   # no collectives below 64 KB before finish()
        self.n = flat_grad.numel()
        self._hi = self.n          # everything in [_hi, n) has been launched
        self._works: List = []
        self._side = streams.shared_stream(flat_grad.device, streams.EXCHANGE) if flat_grad.is_cuda else None
        # SEGMI_GRADSYNC_FORCE=1: issue the collectives even with one rank (a sum over one rank is the
        # identity) -- lets a one-GPU box run the whole RCCL path: side stream, events, Work.wait()
        self._force = os.environ.get("SEGMI_GRADSYNC_FORCE") == "1" and dist.is_initialized()
        self.exposed_events = []

    @property
    def grad_scale(self) -> float:
        return 1.0 / self.world

    def start(self):
        self._hi = self.n
        self._works = []

    def _launch(self, lo: int, hi: int, after=()):
        if hi <= lo:
            return
        chunk = self.flat_grad[lo:hi]
        if self._side is not None:
            ev = torch.cuda.Event()
            ev.record()                      # gradients of the chunk are complete here ...
            with torch.cuda.stream(self._side):
                self._side.wait_event(ev)
                for e in after:              # ... and once these (other-stream) events are reached
                    self._side.wait_event(e)
                self._works.append(dist.all_reduce(chunk, op=dist.ReduceOp.SUM, group=self.group,
                                                   async_op=True))
        else:
            self._works.append(dist.all_reduce(chunk, op=dist.ReduceOp.SUM, group=self.group,
                                               async_op=True))

    def ready(self, lo: int, after=()):
        """gradients at offsets >= lo are final once the current stream's present point and the
        events in ``after`` (work of other streams, e.g. the weight-gradient stream) are reached"""
        if self.world == 1 and not self._force:
            return
        while self._hi - lo >= self.bucket_elems:
            self._launch(self._hi - self.bucket_elems, self._hi, after)
            self._hi -= self.bucket_elems
        # Tail rule: once less than one bucket of the arena is still being computed, do not hold final
        # gradients back for a full bucket -- send what is final now (the deep down-path layers, final
        # ~0.3 ms before the first layer's), so that finish(), whose collective is exposed after the
        # backward, only carries the first layers (a few 10 KB instead of up to a whole bucket).
        if lo < self.bucket_elems and self._hi - lo >= self.tail_min_elems:
            self._launch(lo, self._hi, after)
            self._hi = lo

    # measure = True: bracket finish() with HIP events on the compute stream.  The elapsed time is the
    # part of the exchange the step actually waits for after its backward has ended (the last bucket's
    # collective + whatever earlier buckets have not finished) -- the "exposed" all-reduce time.
    measure = False

    def finish(self, after=()):
        if self.world == 1 and not self._force:
            return
        timed = self.measure and self._side is not None
        if timed:
            e0 = torch.cuda.Event(enable_timing=True)
            e0.record()
        self._launch(0, self._hi, after)
        self._hi = 0
        for w in self._works:
            w.wait()
        self._works = []
        if timed:
            e1 = torch.cuda.Event(enable_timing=True)
            e1.record()
            self.exposed_events.append((e0, e1))

    exposed_events: List = []

    def exposed_ms(self) -> List[float]:
        """per step: time the compute stream spent in finish() (call after a synchronize)"""
        return [a.elapsed_time(b) for a, b in self.exposed_events]


def broadcast_buffers(module: torch.nn.Module, src: int = 0, group=None):
    """DDP-default behaviour of the reference: BN running statistics follow rank 0."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return
    for b in module.buffers():
        dist.broadcast(b.data, src=src, group=group)
