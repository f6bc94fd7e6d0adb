#!/usr/bin/env python3
"""Probe: does replaying the frame's 155 launches as one HIP graph (captured through torch.cuda.CUDAGraph) beat
enqueueing them one by one?  The stream path is GPU-bound (the host runs ahead), so only inter-kernel gaps can go."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import render_in_between_amd as rib
from render_in_between_amd import synth

DTYPE = sys.argv[1] if len(sys.argv) > 1 else "f32"          # f32 | bf16 | f32x3
cfg = rib.hsm_gen_config(); spec = rib.GenSpec.from_cfg(cfg)
G = rib.Generator(cfg, compute_dtype=DTYPE).eval(); G.load_state_dict(synth.make_state_dict(spec, 0))
print("dtype", DTYPE)
label, fake, prev = [t.cuda() for t in synth.make_inputs(spec, 1, 512, 512, 0)]


def step():
    img, mask = G(label, None, fake, prev)
    return G.blend(img, mask, fake)


for _ in range(10):
    step()
torch.cuda.synchronize()


def timeit(fn, n=50):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


print("stream launches: %.3f ms/frame" % timeit(step))
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    for _ in range(3):
        step()
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g, stream=s):
    out = step()
torch.cuda.synchronize()
ref = step()
g.replay(); torch.cuda.synchronize()
print("graph output equals stream output:", bool(torch.equal(out, ref)))
print("graph replay   : %.3f ms/frame" % timeit(g.replay))
print("stream launches: %.3f ms/frame" % timeit(step))
