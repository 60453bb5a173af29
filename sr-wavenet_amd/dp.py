"""Data-parallel glue: one process per GPU, gradients summed with one all-reduce of the flat fp32
buffer (RCCL over xGMI when the backend is "nccl"; gloo in the CPU tests).

Semantics pinned by the reference's loss reductions: both WaveNet losses are means over
the local batch (model.py:29 and the per-sample mu-law CE), so with equal shards the global gradient is
the mean of the shard gradients: all-reduce SUM, then the optimizer multiplies by 1/world.
"""
from __future__ import annotations

import os
from typing import Optional

import torch
import torch.distributed as dist


def init_from_env(backend: Optional[str] = None):
    """Reads RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* (torch.distributed.run); returns (rank, local, world)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        backend = backend or ("nccl" if torch.cuda.is_available() else "gloo")
        kw = {}
        if backend == "nccl":
            kw["device_id"] = torch.device("cuda", local)
        dist.init_process_group(backend, rank=rank, world_size=world, **kw)
    return rank, local, world


def world_size(group=None) -> int:
    return dist.get_world_size(group) if (dist.is_available() and dist.is_initialized()) else 1


def allreduce_sum_(flat: torch.Tensor, group=None, async_op: bool = False):
    """In-place SUM of the flat gradient buffer over ranks (no-op for a single process)."""
    if world_size(group) > 1 or (dist.is_available() and dist.is_initialized() and os.environ.get("SRWN_FORCE_DIST") == "1"):
        return dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group, async_op=async_op)
    return None


def grad_scale(group=None) -> float:
    """Factor the optimizer applies to the summed gradient."""
    return 1.0 / world_size(group)


def shard_batch(global_batch: int, rank: int, world: int):
    """Contiguous equal shards of the global minibatch; raises if it does not divide."""
    if global_batch % world:
        raise ValueError("global batch %d is not divisible by %d ranks" % (global_batch, world))
    per = global_batch // world
    return slice(rank * per, (rank + 1) * per)
