"""One process per GPU: what the path needs from `torch.distributed` (RCCL over xGMI), nothing more.

The reference is single-GPU (no DP/DDP anywhere); these are the builder's
choices for BASELINE.json's multi-GPU configs:

* **Inference** (configs 2, 5) shards by WHOLE harness batches (16 consecutive clips of one
  sub-video): the commit score is a per-batch mean (reference unet.py:310), so a batch is
  never split across GPUs and never re-batched.  No data-path collective; per-frame records
  are gathered at the end (`gather_records`).
* **Training** (configs 3, 4) is plain data parallelism: every rank holds the full 25 M-parameter
  generator, runs the HIP forward/backward on its local clips and averages the gradients with
  a bucketed all-reduce.  `BucketedGradReducer` is driven by the training engine as each stage
  of the hand-scheduled backward finishes (decoders -> bridge -> memory -> encoders), so the
  RCCL traffic of early buckets overlaps the remaining backward kernels.  100 MB of fp32
  gradients per step is ~1.2 ms on one xGMI link; buckets of ~25 MB keep RCCL's ring/tree
  protocols in their bandwidth regime without delaying the first launch.
  BatchNorm statistics and the EMA codebook stay per-rank (stock DDP semantics); buffers are
  made identical across ranks with `broadcast_state` (rank 0 wins) at start and on demand.
"""
from __future__ import annotations

import os
from typing import Dict, Iterable, List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist


def init_distributed(backend: Optional[str] = None) -> Tuple[int, int, torch.device]:
    """RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from the environment (torchrun); backend
    "nccl" (= RCCL on ROCm) when a GPU is visible, else "gloo"."""
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    use_gpu = torch.cuda.is_available()
    dev = torch.device("cuda", local) if use_gpu else torch.device("cpu")
    if use_gpu:
        torch.cuda.set_device(dev)
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        backend = backend or ("nccl" if use_gpu else "gloo")
        kw = {"device_id": dev} if backend == "nccl" else {}
        dist.init_process_group(backend, rank=rank, world_size=world, **kw)
    return rank, world, dev


def shard_batches(n_batches: int, rank: int, world: int) -> List[int]:
    """indices of the harness batches this rank evaluates (contiguous blocks, so that a
    sub-video's batches mostly stay on one GPU); every batch is owned by exactly one rank"""
    base, extra = divmod(n_batches, world)
    start = rank * base + min(rank, extra)
    return list(range(start, start + base + (1 if rank < extra else 0)))


def gather_records(local: Dict[int, object], world: int) -> Dict[int, object]:
    """merge {batch index: per-frame records} from all ranks (a few KB; object all-gather)"""
    if world == 1 or not dist.is_initialized():
        return dict(local)
    parts: List[Optional[dict]] = [None] * world
    dist.all_gather_object(parts, local)
    out: Dict[int, object] = {}
    for p in parts:
        out.update(p)
    return out


def broadcast_state(module: torch.nn.Module, src: int = 0) -> None:
    """make parameters and buffers (BN running stats, codebook) identical on every rank"""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return
    with torch.no_grad():
        for t in list(module.parameters()) + list(module.buffers()):
            dist.broadcast(t.data, src)
    if hasattr(module, "_param_epoch"):
        module._param_epoch += 1


class BucketedGradReducer:
    """Average gradients across ranks in ~`bucket_mb` MiB buckets, launched as they become ready.

    `push(tensors)` may be called many times during one backward; a bucket is flattened and
    all-reduced asynchronously as soon as it is full.  `finish()` flushes the tail, waits and
    scatters the averaged values back IN PLACE into the pushed tensors.  `force`: take the collective path with ONE
    rank too (an initialised process group is then required) - the one-GPU proof that the communicator, the
    asynchronous all-reduce and its ordering against the raw-pointer kernels on torch's current stream work."""

    def __init__(self, bucket_mb: float = 25.0, group=None, force: bool = False):
        self.bucket_bytes = int(bucket_mb * (1 << 20))
        self.group = group
        self.force = bool(force)
        self._flats: Dict[int, torch.Tensor] = {}
        self._pending: List[torch.Tensor] = []
        self._pending_bytes = 0
        self._inflight: List[Tuple[torch.Tensor, List[torch.Tensor], object]] = []
        self.buckets_launched = 0
        # measurement (bench.py --mode train --gpus N): with `time_finish` every `finish()` is bracketed by HIP events on
        # the compute stream - what the step pays for the collectives that did NOT hide behind the backward (the wait
        # for the last buckets, the averaging and the scatter back) - collected in `finish_events`
        self.time_finish = False
        self.finish_events: List[Tuple[torch.cuda.Event, torch.cuda.Event]] = []
        self.last_step_buckets = 0

    @property
    def world(self) -> int:
        return dist.get_world_size(self.group) if dist.is_initialized() else 1

    @property
    def active(self) -> bool:
        return self.world > 1 or (self.force and dist.is_initialized())

    def push(self, tensors: Iterable[torch.Tensor]) -> None:
        if not self.active:
            return
        for t in tensors:
            self._pending.append(t)
            self._pending_bytes += t.numel() * t.element_size()
            if self._pending_bytes >= self.bucket_bytes:
                self._launch()

    def _flat_for(self, slot: int, members: List[torch.Tensor]) -> torch.Tensor:
        """the persistent flat buffer of the `slot`-th bucket of a step: the bucket composition is the same every step
        (the backward pushes the same tensors in the same order), so the buffer is allocated once and only re-made
        when the total size or dtype / device changes - no `torch.cat` allocation per bucket per step"""
        n = sum(t.numel() for t in members)
        flat = self._flats.get(slot)
        if flat is None or flat.numel() != n or flat.dtype != members[0].dtype or flat.device != members[0].device:
            flat = torch.empty(n, dtype=members[0].dtype, device=members[0].device)
            self._flats[slot] = flat
        return flat

    def _launch(self) -> None:
        if not self._pending:
            return
        members = self._pending
        flat = self._flat_for(len(self._inflight), members)
        views, off = [], 0
        for t in members:
            n = t.numel()
            views.append(flat[off:off + n].view_as(t))
            off += n
        torch._foreach_copy_(views, [t.detach() for t in members])          # one fused gather into the bucket
        work = dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        self._inflight.append((views, members, work))
        self._pending, self._pending_bytes = [], 0
        self.buckets_launched += 1

    def finish(self) -> None:
        if not self.active:
            return
        self._launch()
        inv = 1.0 / self.world
        self.last_step_buckets = len(self._inflight)
        ev = None
        if self.time_finish and self._inflight and self._inflight[0][0][0].is_cuda:
            ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            ev[0].record()
        for views, members, work in self._inflight:
            work.wait()
            torch._foreach_mul_(views, inv)                                  # average in the bucket ...
            torch._foreach_copy_([t.detach() for t in members], views)       # ... and scatter back in place
        if ev is not None:
            ev[1].record()
            self.finish_events.append(ev)
        self._inflight = []


def attach_reducer(module: torch.nn.Module, reducer: Optional[BucketedGradReducer]) -> None:
    """let the model's training engine feed `reducer` stage by stage during backward"""
    object.__setattr__(module, "_grad_reducer", reducer)


def sync_statistics(module: torch.nn.Module, enabled: bool = True, group=None, force: bool = False) -> None:
    """Large-batch-exact data parallelism: BatchNorm batch statistics (forward sums and the
    two backward sums per layer, [2C] floats each) and the per-slot counts / feature sums of the EMA codebook
    update are all-reduced across ranks, so N ranks x B clips reproduce ONE step on N*B clips (running statistics
    and codebook identical on every rank).  Ranks must hold equal batch sizes.  Off by default: the stock
    behaviour keeps statistics per rank.  `force`: take the collective path with a world of one as well (one-GPU test of
    the 33 collectives of a step over RCCL)."""
    object.__setattr__(module, "_sync_stats", (bool(enabled), group, bool(force)))
