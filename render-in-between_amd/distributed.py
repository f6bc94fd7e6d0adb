"""Multi-GPU host logic: one process per GPU, torch.distributed over RCCL/xGMI.

The path shards with no exchange step: segments / clips are
independent (evaluator.py:240-244 resets the recurrence at every key frame), so
the only collective is ONE broadcast of the folded weight blob at start-up.
"""
from __future__ import annotations

import os
import socket
import subprocess
import sys
import time
from typing import List, Sequence

import torch
import torch.distributed as dist


def free_port() -> int:
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def launch_command(script: str, argv: Sequence[str], nproc: int, port: int = None) -> List[str]:
    """The command that starts `nproc` ranks of `script` on this node, one per GPU, exactly as the driver does it
    (python -m torch.distributed.run, rendezvous on 127.0.0.1)."""
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(int(nproc)),
            "--master-addr", "127.0.0.1", "--master-port", str(port or free_port()), script] + list(argv)


def self_launch(script: str, argv: Sequence[str], nproc: int) -> int:
    """Called by a program that was asked for `nproc` > 1 GPUs but is not a rank yet (WORLD_SIZE unset): start
    the ranks as CHILD processes and return their exit code.  Must run before the caller touches the GPU: the ranks
    are fresh processes, nothing re-executes after HIP has been initialised.  Rank 0's stdout is the caller's."""
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC only on this host driver (RCCL needs it)
    return subprocess.call(launch_command(script, argv, nproc), env=env)


def is_rank_process() -> bool:
    return "WORLD_SIZE" in os.environ and "RANK" in os.environ


def gather_rows(row: torch.Tensor, rank: int, world: int) -> torch.Tensor:
    """[world, *row.shape] with every rank's row, on every rank.  One all-reduce(SUM) of a buffer that is zero except
    for the caller's own row: exact (x + 0), and available for device tensors on both RCCL and gloo (all_gather of
    device tensors is not, on gloo).  Control plane only (timings, checksums, the frames of the replica check)."""
    buf = torch.zeros((world,) + tuple(row.shape), dtype=row.dtype, device=row.device)
    buf[rank] = row
    if world > 1:
        dist.all_reduce(buf, op=dist.ReduceOp.SUM)
    return buf


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
