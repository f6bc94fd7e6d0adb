#!/usr/bin/env python3
"""Debug aid: pin ONE launch at a time to each of its fitting DMA-staged k_igemm variants and print the frame's deviation."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import render_in_between_amd as rib
from render_in_between_amd import synth, _native
cfg = rib.hsm_gen_config(); spec = rib.GenSpec.from_cfg(cfg); sd = synth.make_state_dict(spec, 0)
lib = _native.lib()
g12 = (C.c_int * 12)()
geoms = {i: list(g12) for i in range(lib.rib_num_variants()) if lib.rib_variant_info(i, g12) == 0 and g12[11] == 100}
B, H, W = 1, 128, 128
label, fake, prev = synth.make_inputs(spec, B, H, W, 31)
G0 = rib.Generator(cfg, use_tuning=False).eval(); G0.load_state_dict(sd)
i0, m0 = [t.clone() for t in G0(label, None, fake, prev)]
G1 = rib.Generator(cfg, use_tuning=False).eval(); G1.load_state_dict(sd)
names = [x["name"] for x in G1.launch_info(B, H, W) if x["class"] in (0, 1) and "tile " in x["tile"] and "gemm" not in x["tile"] and "wino" not in x["tile"]]
only = sys.argv[1] if len(sys.argv) > 1 else None
for nm in names:
    if only and only not in nm: continue
    for vi, g in geoms.items():
        lib.rib_set_choice(G1._h, B, H, W, nm.encode(), vi, 1)
        if lib.rib_workspace_bytes(G1._h, B, H, W) == 0:
            lib.rib_set_choice(G1._h, B, H, W, nm.encode(), -1, 1); continue
        G1._ws.clear()
        i1, m1 = G1(label, None, fake, prev)
        info = [x["tile"] for x in G1.launch_info(B, H, W) if x["name"] == nm][0]
        print("%-44s %-28s img %.2e mask %.2e" % (nm[-44:], str(g[:9]), float((i1 - i0).abs().max()), float((m1 - m0).abs().max())), flush=True)
        lib.rib_set_choice(G1._h, B, H, W, nm.encode(), -1, 1)
