#!/usr/bin/env python3
"""Time single launches of the frame plan in isolation (rib_time_op: the op's own sub-plan, `iters` back-to-back launches).

    python tools/time_ops.py [--size 512 | --height H --width W] [--batch B] [--dtype f32] name-substring ...
"""
import argparse, ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import render_in_between_amd as rib
from render_in_between_amd import _native, synth


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("names", nargs="+")
    ap.add_argument("--size", type=int, default=512)
    ap.add_argument("--height", type=int, default=0)
    ap.add_argument("--width", type=int, default=0)
    ap.add_argument("--batch", type=int, default=1)
    ap.add_argument("--iters", type=int, default=50)
    ap.add_argument("--dtype", default="f32")
    a = ap.parse_args()
    B, H, W = a.batch, a.height or a.size, a.width or a.size
    lib = _native.lib()
    cfg = rib.hsm_gen_config(); spec = rib.GenSpec.from_cfg(cfg)
    G = rib.Generator(cfg, compute_dtype=a.dtype).eval()
    G.load_state_dict(synth.make_state_dict(spec, 0))
    label, fake, prev = [t.cuda() for t in synth.make_inputs(spec, B, H, W, 0)]
    img, mask, fuse = G.forward_blend(label, None, fake, prev)
    ws = G._workspace(B, H, W)
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    usec = C.c_double()
    total = 0.0
    for info in G.launch_info(B, H, W):
        if not any(n in info["name"] for n in a.names):
            continue
        best = 1e9
        for _ in range(3):
            rc = lib.rib_time_op(G._h, B, H, W, info["name"].encode(), *[C.c_void_p(t.data_ptr()) for t in (label, fake, prev, img, mask)],
                                 C.c_void_p(ws.data_ptr()), ws.numel(), a.iters, st, C.byref(usec))
            assert rc == 0, lib.rib_last_error(G._h)
            best = min(best, usec.value)
        total += best
        print("%-50s %8.1f us  %s" % (info["name"], best, info["tile"][:70]))
    print("%-50s %8.1f us" % ("total", total))


if __name__ == "__main__":
    main()
