#!/usr/bin/env python3
"""Thread sweep of the CPU oracle at 512x512 (choosing the cpu_baseline thread count)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import render_in_between_amd as rib
from render_in_between_amd import synth
from oracle import generator_ref
spec = rib.GenSpec.from_cfg(rib.hsm_gen_config())
R = generator_ref.RefGenerator(spec, synth.make_state_dict(spec, 0))
l, f, p = synth.make_inputs(spec, 1, 512, 512, 0)
for nt in [int(x) for x in sys.argv[1:]] or [8, 16, 32, 64]:
    torch.set_num_threads(nt)
    R(l, None, f, p)
    t = time.perf_counter(); R(l, None, f, p); R(l, None, f, p); dt = (time.perf_counter() - t) / 2
    print("threads %3d: %.3f s/frame (%.2f fps)" % (nt, dt, 1 / dt), flush=True)
