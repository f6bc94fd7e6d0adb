#!/usr/bin/env python3
"""RCCL path of the multi-GPU code on ONE GPU: a world of one rank goes through exactly the calls the N-rank bench makes
(init_process_group(nccl, device_id), agree_or_raise, the weight-blob broadcast, barrier, the max-over-ranks all-reduce).
Two ranks on one GPU are refused by RCCL (duplicate device), so this is as far as a 1-GPU box goes; the N-rank logic itself is
covered by the gloo tests.

    MASTER_ADDR=127.0.0.1 MASTER_PORT=29533 RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 python3 tools/rccl_smoke.py
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch
import torch.distributed as dist

import render_in_between_amd as rib
from render_in_between_amd import distributed as ribdist, synth


def main():
    for k, v in (("MASTER_ADDR", "127.0.0.1"), ("MASTER_PORT", "29533"), ("RANK", "0"), ("LOCAL_RANK", "0"), ("WORLD_SIZE", "1")):
        os.environ.setdefault(k, v)
    idx = ribdist.rank_device_index()
    torch.cuda.set_device(idx)
    dev = torch.device("cuda", idx)
    t0 = time.perf_counter()
    ribdist.init_process_group("nccl", dev)
    assert dist.get_backend() == "nccl" and dist.get_world_size() == 1
    ribdist.agree_or_raise(True, "rank 0 could not load the checkpoint", device=dev)
    cfg = rib.hsm_gen_config()
    spec = rib.GenSpec.from_cfg(cfg)
    G = rib.Generator(cfg, device=dev).eval()
    G.load_state_dict(synth.make_state_dict(spec, 0))
    before = ribdist.blob_checksum(G.export_weights())
    ms = ribdist.broadcast_weights(G, src=0)
    after = ribdist.blob_checksum(G.export_weights())
    assert before == after, (before, after)
    dist.barrier()
    t = torch.tensor([1.25], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    assert float(t.item()) == 1.25
    label, fake, prev = [x.to(dev) for x in synth.make_inputs(spec, 1, 128, 128, 0)]
    img, mask, fuse = G.forward_blend(label, None, fake, prev)
    torch.cuda.synchronize()
    assert torch.isfinite(fuse).all()
    print("rccl smoke ok: backend %s, device %s, broadcast of %.1f MB in %.2f ms, %.1f s total"
          % (dist.get_backend(), ribdist.device_identity(idx), G.export_weights().numel() * G.export_weights().element_size() / 1e6, ms, time.perf_counter() - t0))
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
