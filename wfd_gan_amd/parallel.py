"""Data-parallel helpers (one process per GPU, torch.distributed; backend "nccl" = RCCL over xGMI on the GPU box,
"gloo" in the CPU tests).  The training path shards by batch: every rank runs the same plans on its own rows and
the only exchange is the gradient all-reduce of the two flat gradient buffers (22.8 MB critic per iteration,
23.2 MB generator every n_critic-th iteration).  Sampling needs no collective."""
from __future__ import annotations

import os
from typing import List, Optional, Tuple

import torch
import torch.distributed as dist


def init_from_env(backend: Optional[str] = None) -> Tuple[int, int, int]:
    """(rank, world, local_rank) from the torchrun environment; initialises the default process group if world > 1."""
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def allreduce_sum_(flat: torch.Tensor, group=None) -> torch.Tensor:
    """In-place sum of a flat gradient buffer over all ranks (the optimizer applies the 1/world factor)."""
    dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
    return flat


class GradSync:
    """Bucketed gradient all-reduce that overlaps with the rest of the backward: ``start(bucket)`` launches an
    asynchronous sum over a slice of the flat gradient buffer as soon as that slice is final, ``finish()`` makes
    the current stream wait for every bucket before the optimizer kernel reads the buffer."""

    def __init__(self, group=None):
        self.group = group
        self.pending = []

    def start(self, bucket: torch.Tensor):
        self.pending.append(dist.all_reduce(bucket, op=dist.ReduceOp.SUM, group=self.group, async_op=True))

    def finish(self):
        for w in self.pending:
            w.wait()
        self.pending.clear()


def shard_rows(n_rows: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous [begin, end) shard of n_rows for this rank; the first n_rows % world ranks get one extra row."""
    base, rem = divmod(n_rows, world)
    begin = rank * base + min(rank, rem)
    return begin, begin + base + (1 if rank < rem else 0)


def max_over_ranks(value: float, device) -> float:
    """Timing convention of bench.py: the slowest rank defines the step time."""
    if not (dist.is_available() and dist.is_initialized()):
        return value
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(value: float, device) -> float:
    if not (dist.is_available() and dist.is_initialized()):
        return value
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())


def broadcast_state_(tensors: List[torch.Tensor], src: int = 0, group=None) -> None:
    """Make replicas identical at start-up (parameters, optimizer state, BN buffers)."""
    if dist.is_available() and dist.is_initialized():
        for t in tensors:
            dist.broadcast(t, src=src, group=group)
