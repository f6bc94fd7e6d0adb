#!/usr/bin/env python3
"""Debug aid for the bf16 precision mode (argv[1]): per-tap error vs the fp32 CPU oracle (in plan order), then the
frame error at a few sizes.  Run on the GPU box."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import render_in_between_amd as rib
from render_in_between_amd import synth
from oracle import generator_ref

MID = dict(num_filters=16, max_num_filters=64, mask=dict(num_filters=32, max_num_filters=64), embed=dict(num_filters=32, max_num_filters=64))
for name, cfg, sizes in (("mid", rib.hsm_gen_config(**MID), [(1, 64, 64)]), ("full", rib.hsm_gen_config(), [(1, 64, 64), (2, 48, 80), (1, 256, 256)])):
    spec = rib.GenSpec.from_cfg(cfg)
    sd = synth.make_state_dict(spec, 7)
    G = rib.Generator(cfg, compute_dtype=(sys.argv[1] if len(sys.argv) > 1 else "bf16")).eval()
    G.load_state_dict(sd)
    R = generator_ref.RefGenerator(spec, sd)
    for (B, H, W) in sizes:
        label, fake, prev = synth.make_inputs(spec, B, H, W, 7)
        G.enable_taps()
        img, mask = G(label, None, fake, prev)
        torch.cuda.synchronize()
        taps = G.read_taps(B, H, W)
        otaps = {}
        oi, om = R(label, None, fake, prev, taps=otaps)
        print("== %s %s: img %.3e mask %.3e (mean %.3e %.3e)" % (name, (B, H, W), float((img.cpu() - oi).abs().max()), float((mask.cpu() - om).abs().max()),
                                                                  float((img.cpu() - oi).abs().mean()), float((mask.cpu() - om).abs().mean())))
        if (B, H, W) == (1, 64, 64):
            for k, v in taps.items():
                ref = otaps[k]
                print("   %-28s rel max %.3e  rel mean %.3e" % (k, float((v - ref).abs().max()) / max(1.0, float(ref.abs().max())), float((v - ref).abs().mean()) / max(1e-6, float(ref.abs().mean()))))
        G.enable_taps(False)
