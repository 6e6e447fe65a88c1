"""Multi-GPU use of the path: one process per GPU, batch sharded across ranks, outputs gathered over RCCL.

Clips are independent (the reference has no cross-sample op), so every rank holds a full weight replica and runs
`encode_audio` / `decode_audio` on its contiguous slice of the batch.  The only exchange step is the final
all-gather of quantiser indices and waveforms (backend "nccl" = RCCL over xGMI on ROCm; "gloo" on CPU in the tests).
"""
from __future__ import annotations

from typing import Optional

import torch
import torch.distributed as dist


def shard_range(total: int, rank: int, world: int) -> tuple[int, int]:
    """Contiguous slice [start, stop) of `total` clips owned by `rank`; the first `total % world` ranks get one more."""
    if not (0 <= rank < world):
        raise ValueError(f"rank {rank} outside world of {world}")
    base, extra = divmod(total, world)
    start = rank * base + min(rank, extra)
    return start, start + base + (1 if rank < extra else 0)


def shard_batch(batch: torch.Tensor, rank: Optional[int] = None, world: Optional[int] = None) -> torch.Tensor:
    rank = dist.get_rank() if rank is None else rank
    world = dist.get_world_size() if world is None else world
    start, stop = shard_range(batch.shape[0], rank, world)
    return batch[start:stop]


def gather_batch(local: torch.Tensor, total: int, group=None, force_ragged: bool = False) -> torch.Tensor:
    """All-gather per-rank slices (as produced by `shard_range`) back into the full batch, on every rank.
    `force_ragged` takes the padded path (the one a batch that does not divide by the world size needs) whatever the sizes,
    also in a world of one: it is how that path is exercised on a single GPU (tests)."""
    world = dist.get_world_size(group)
    if world == 1 and not force_ragged:
        return local
    base, extra = divmod(total, world)
    if extra == 0 and not force_ragged:  # equal shards: one collective straight into the output
        out = torch.empty((total,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
        dist.all_gather_into_tensor(out, local.contiguous(), group=group)
        return out
    # ragged: pad every shard to base + 1 rows, gather, drop the padding
    rows = base + 1
    padded = torch.zeros((rows,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    padded[: local.shape[0]] = local
    out = torch.empty((world * rows,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(out, padded, group=group)
    parts = []
    for r in range(world):
        n = base + (1 if r < extra else 0)
        parts.append(out[r * rows: r * rows + n])
    return torch.cat(parts, dim=0)


class PendingGather:
    """Handle of an all-gather in flight on the process group's own stream (RCCL runs it beside the compute stream).
    `wait()` makes the CURRENT stream wait for it and returns the gathered tensor; the shard stays referenced until then."""

    def __init__(self, out: torch.Tensor, work=None, keep=None):
        self.out, self.work, self.keep = out, work, keep

    def wait(self) -> torch.Tensor:
        if self.work is not None:
            self.work.wait()
            self.work = self.keep = None
        return self.out


def gather_batch_async(local: torch.Tensor, total: int, group=None, force: bool = False) -> PendingGather:
    """`gather_batch` without blocking the compute stream: the collective is queued behind the work already on the current
    stream and runs concurrently with whatever is launched next (e.g. the next batch's encode).  Equal shards only; ragged
    batches fall back to the blocking path.  `force` issues the collective even in a world of one (tests)."""
    world = dist.get_world_size(group)
    if world == 1 and not force:
        return PendingGather(local)
    if total % world:
        return PendingGather(gather_batch(local, total, group))
    src = local.contiguous()
    out = torch.empty((total,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    work = dist.all_gather_into_tensor(out, src, group=group, async_op=True)
    return PendingGather(out, work, src)


class PendingGathers:
    """The overlap loop of a stream of batches: each step pushes the gathers it has just issued and retires the ones of the
    step before, so an exchange is in flight while the next batch is being encoded and at most one step's outputs wait.
    `drain()` retires whatever is left (end of the stream, or before timing stops).  `results` keeps the last retired
    step's gathered tensors."""

    def __init__(self):
        self._steps: list[tuple[PendingGather, ...]] = []
        self.results: tuple[torch.Tensor, ...] = ()
        self.retired = 0

    def push(self, *handles: PendingGather) -> None:
        self.drain()
        self._steps.append(tuple(handles))

    def drain(self) -> None:
        while self._steps:
            self.results = tuple(h.wait() for h in self._steps.pop(0))
            self.retired += 1

    def __len__(self) -> int:
        return len(self._steps)


class ShardedCodec:
    """Runs a codec (anything with `encode_audio` / `decode_audio`, e.g. `l3ac_amd.L3AC`) on this rank's slice of a
    batch that every rank holds, and returns the gathered full-batch outputs on every rank."""

    def __init__(self, codec, group=None):
        self.codec = codec
        self.group = group

    def encode_decode(self, audio: torch.Tensor):
        total = audio.shape[0]
        local = shard_batch(audio, dist.get_rank(self.group), dist.get_world_size(self.group))
        q_feature, ind = self.codec.encode_audio(local)
        wave = self.codec.decode_audio(q_feature)
        return gather_batch(ind["indices"], total, self.group), gather_batch(wave, total, self.group)
