"""Multi-GPU host logic: one process per GPU, torch.distributed over RCCL/xGMI.

The path shards with no exchange step: segments / clips are
independent (evaluator.py:240-244 resets the recurrence at every key frame), so
the only collective is ONE broadcast of the folded weight blob at start-up.
"""
from __future__ import annotations

import contextlib
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
    """The command that starts `nproc` ranks of `script` on this node, one per GPU (python -m torch.distributed.run,
    rendezvous on 127.0.0.1).  With a port: exactly the driver's form (--master-addr / --master-port).  Without one:
    --standalone, where the launcher's own TCP store binds a free port and keeps it - no window between picking a port
    and binding it in which another process could take it."""
    head = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(int(nproc))]
    if port is None:
        rdzv = ["--standalone", "--local-addr", "127.0.0.1"]
    else:
        rdzv = ["--master-addr", "127.0.0.1", "--master-port", str(int(port))]
    return head + rdzv + [script] + list(argv)


def rank_device_index(environ=None) -> int:
    """HIP device of this rank: LOCAL_RANK, one process per GPU (torch.distributed.run gives the N ranks of a node the
    LOCAL_RANKs 0..N-1: N distinct devices).  RIB_BENCH_DEVICE overrides it - only to rehearse several ranks on a
    one-GPU box."""
    env = os.environ if environ is None else environ
    return int(env.get("RIB_BENCH_DEVICE", env.get("LOCAL_RANK", "0")))


def device_identity(index: int) -> str:
    """'cuda:<i> <name> pci <bus id>' of a visible device, for the per-rank line of the multi-GPU bench."""
    try:
        p = torch.cuda.get_device_properties(index)
        bus = getattr(p, "pci_bus_id", None)
        dom = getattr(p, "pci_domain_id", 0)
        dev = getattr(p, "pci_device_id", 0)
        where = "%04x:%02x:%02x" % (dom, bus, dev) if bus is not None else "?"
        return "cuda:%d %s pci %s" % (index, p.name, where)
    except Exception as e:                                          # noqa: BLE001  (identity is informational)
        return "cuda:%d (%s)" % (index, type(e).__name__)


def self_launch(script: str, argv: Sequence[str], nproc: int) -> int:
    """Called by a program that was asked for `nproc` > 1 GPUs but is not a rank yet (WORLD_SIZE unset): start
    the ranks as CHILD processes and return their exit code.  Must run before the caller touches the GPU: the ranks
    are fresh processes, nothing re-executes after HIP has been initialised.  Rank 0's stdout is the caller's."""
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC only on this host driver (RCCL needs it)
    rc = subprocess.call(launch_command(script, argv, nproc), env=env)
    if rc != 0:
        print("[rib launcher] %d ranks of %s ended with exit code %d (HSA_ENABLE_IPC_MODE_LEGACY=%s); the failing rank's own line starts "
              "with '[rib rank' above" % (nproc, os.path.basename(script), rc, env["HSA_ENABLE_IPC_MODE_LEGACY"]), file=sys.stderr, flush=True)
    return rc


@contextlib.contextmanager
def startup_phase(what: str, device_index: int = None):
    """Makes a failure of the multi-GPU start-up self-diagnosing (RCCL has only ever run in a world of
    one rank on the build's boxes, so the first real 8-GPU run must say what went wrong by itself).  Any exception inside
    the block - process-group creation, the first communicator, the weight broadcast - is reported on stderr as ONE line
    that names the rank, its device, the backend, the RCCL / HIP error text and the IPC mode in force, and is then re-raised:
    the rank exits non-zero, torch.distributed.run ends the other ranks, and self_launch (a parent that never touched the
    GPU) hands the code back."""
    try:
        yield
    except BaseException as e:      # noqa: BLE001 (reported and re-raised)
        print(failure_line(what, e, device_index), file=sys.stderr, flush=True)
        raise


def failure_line(what: str, exc: BaseException, device_index: int = None) -> str:
    env = os.environ
    idx = rank_device_index() if device_index is None else device_index
    backend = dist.get_backend() if (dist.is_available() and dist.is_initialized()) else env.get("RIB_DIST_BACKEND", "nccl (RCCL)" if torch.cuda.is_available() else "gloo")
    ident = device_identity(idx) if torch.cuda.is_available() else "cuda:%d (no GPU visible)" % idx
    msg = " ".join(str(exc).split())[:400] or type(exc).__name__
    return ("[rib rank %s/%s local_rank %s] FAILED during %s | device %s | backend %s | HSA_ENABLE_IPC_MODE_LEGACY=%s (0 = dmabuf IPC, what this "
            "host driver supports) NCCL_DEBUG=%s MASTER=%s:%s | %s: %s"
            % (env.get("RANK", "0"), env.get("WORLD_SIZE", "1"), env.get("LOCAL_RANK", "0"), what, ident, backend,
               env.get("HSA_ENABLE_IPC_MODE_LEGACY", "unset"), env.get("NCCL_DEBUG", "unset"), env.get("MASTER_ADDR", "?"), env.get("MASTER_PORT", "?"),
               type(exc).__name__, msg))


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


def init_process_group(backend: str = None, device: torch.device = None):
    """`device`: this rank's GPU.  Handed to init_process_group as device_id with the RCCL backend, so that the
    communicator is bound to it up front (eager init, no guessing from the rank number at the first collective)."""
    if dist.is_initialized():
        return
    if backend is None:
        backend = "nccl" if torch.cuda.is_available() else "gloo"   # "nccl" is RCCL on ROCm
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29517")
    kw = {}
    if backend == "nccl" and device is not None and torch.device(device).type == "cuda":
        kw["device_id"] = torch.device(device)
    # RCCL (and gloo) print to STDOUT while a communicator comes up - with NCCL_DEBUG=VERSION, which is exported on the GPU
    # boxes, a five-line banner - and stdout belongs to the callers' results (bench.py: ONE JSON line).  The environment is
    # left as the user set it (NCCL_DEBUG, NCCL_DEBUG_FILE); instead file descriptor 1 points at stderr while the group
    # and its first communicator are created (a one-element all-reduce forces the latter), and is restored afterwards.
    with _stdout_to_stderr(), startup_phase("init_process_group(%s) + first communicator" % backend, getattr(kw.get("device_id"), "index", None)):
        dist.init_process_group(backend=backend, **kw)
        t = torch.zeros(1, dtype=torch.int32, device=kw.get("device_id", "cpu"))
        dist.all_reduce(t)
        if t.is_cuda:
            torch.cuda.synchronize(t.device)


@contextlib.contextmanager
def _stdout_to_stderr():
    sys.stdout.flush()
    saved = os.dup(1)
    try:
        os.dup2(2, 1)
        yield
    finally:
        sys.stdout.flush()
        try:
            import ctypes
            ctypes.CDLL(None).fflush(None)      # C stdio buffers of the libraries that printed (a pipe is block-buffered)
        except OSError:
            pass
        os.dup2(saved, 1)
        os.close(saved)


def agree_or_raise(ok: bool, what: str, device=None):
    """Every rank calls this with its own verdict; all ranks raise together when any of them failed (one
    all-reduce(MIN) of a flag).  Used in front of a collective that only one rank prepares - rank 0 reading the
    checkpoint before the weight broadcast - so that a failure there ends the job with ONE clear error instead of
    leaving the other ranks blocked in the collective until the launcher tears them down."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        if not ok:
            raise RuntimeError(what)
        return
    dev = device if (device is not None and dist.get_backend() == "nccl") else "cpu"
    flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device=dev)
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    if int(flag.item()) == 0:
        raise RuntimeError(what if not ok else "another rank failed: " + what)


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
    device blob (one RCCL broadcast: 126 MB in fp32 mode - every handle makes the Winograd-domain filter sets it
    needs on its own device - 188 MB in the 16-bit modes) and adopts it (rib_import_weights checks the blob's header: mode and layout must match).
    Returns the broadcast wall time in ms (synchronised)."""
    rank = dist.get_rank()
    with startup_phase("the weight broadcast (one %s broadcast of the folded blob from rank %d)" % (dist.get_backend(), src), getattr(gen.device, "index", None)):
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
