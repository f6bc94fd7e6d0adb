"""Multi-GPU host logic: one process per GPU, torch.distributed over RCCL/xGMI.

The path shards with no exchange step: segments / clips are
independent (evaluator.py:240-244 resets the recurrence at every key frame), so
the only collective is ONE broadcast of the folded weight blob at start-up.
"""
from __future__ import annotations

import os
import time
from typing import List, Sequence

import torch
import torch.distributed as dist


def init_process_group(backend: str = None):
    if dist.is_initialized():
        return
    if backend is None:
        backend = "nccl" if torch.cuda.is_available() else "gloo"   # "nccl" is RCCL on ROCm
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29517")
    dist.init_process_group(backend=backend)


def shard_units(num_units: int, rank: int, world: int) -> List[int]:
    """Static round-robin partition of independent units of work (segments or
    clips; all units cost the same at fixed H, W)."""
    return list(range(rank, num_units, world))


def broadcast_blob(buf: torch.Tensor, src: int = 0) -> torch.Tensor:
    """The single data-path collective: broadcast one contiguous buffer."""
    dist.broadcast(buf, src=src)
    return buf


def broadcast_weights(gen, src: int = 0) -> float:
    """Rank `src` holds loaded weights; every other rank receives the folded
    device blob (one RCCL broadcast, ~123 MB fp32) and adopts it.  Returns the
    broadcast wall time in ms (synchronised)."""
    rank = dist.get_rank()
    if rank == src:
        buf = gen.export_weights()
    else:
        buf = torch.empty(gen.weights_numel(), dtype=torch.float32, device=gen.device)
    torch.cuda.synchronize(gen.device)
    t0 = time.perf_counter()
    broadcast_blob(buf, src)
    torch.cuda.synchronize(gen.device)
    ms = (time.perf_counter() - t0) * 1e3
    if rank != src:
        gen.import_weights(buf)
        torch.cuda.synchronize(gen.device)
    return ms


def blob_checksum(buf: torch.Tensor) -> int:
    """Order-independent integer checksum of a blob (equal on all ranks after the broadcast)."""
    return int(buf.view(torch.int32).to(torch.int64).sum().item())
