"""The RCCL calls of the multi-GPU path on the one GPU a test box has: a world of one rank goes through
init_process_group(nccl, device_id), the agreement all-reduce, the weight-blob broadcast, barrier and the max-over-ranks
all-reduce (tools/rccl_smoke.py), in a fresh process.  The N-rank logic (sharding, launch command, failure handling, the
device map) is covered by the gloo tests in test_distributed_cpu.py; SURVEY 8(e)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_rccl_world_of_one_runs_the_collectives_of_the_multi_gpu_path():
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29541", RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", NCCL_DEBUG="VERSION")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "rccl_smoke.py")], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "rccl smoke ok: backend nccl" in r.stdout, r.stdout[-2000:]
    # stdout is the callers' result channel (bench.py prints ONE JSON line): RCCL's version banner (NCCL_DEBUG=VERSION is
    # exported on the GPU boxes) must not land there
    assert "RCCL version" not in r.stdout and len(r.stdout.strip().splitlines()) == 1, r.stdout[-2000:]
