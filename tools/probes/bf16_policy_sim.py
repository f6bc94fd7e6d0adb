#!/usr/bin/env python3
"""CPU model of the bf16 storage mode's roundings, to find out WHERE its error comes from before spending fp32 anywhere
(VERDICT r02 item 5).  The oracle's forward (oracle/generator_ref.py) is re-run with torch's F.conv2d / the stored
tensors rounded to bf16 at the places the bf16 kernels round them:

  * every tensor a kernel stores (conv outputs after bias / residual / activation, SPADE outputs, pooled tensors, joins),
  * the prologue's result on its way into LDS (IN affine + LeakyReLU is fp32 arithmetic, then rounded again),
  * the filters of the matrix-core kernels (gamma/beta filters included); fp32 accumulation = exact products of the
    rounded operands summed in fp32.

A policy switches groups of these roundings off; the table shows max / mean |error| of img and mask against the fp32
oracle.  Runs in the build container (no GPU): `python tools/probes/bf16_policy_sim.py [size]`.
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

import torch

import render_in_between_amd as rib
from render_in_between_amd import synth
from oracle import generator_ref as R


from oracle.precision_model import Sim          # noqa: E402  (the model itself lives with the oracle since round 6)


def main():
    size = int(sys.argv[1]) if len(sys.argv) > 1 else 128
    torch.set_num_threads(8)
    cfg = rib.hsm_gen_config()
    spec = rib.GenSpec.from_cfg(cfg)
    sd = synth.make_state_dict(spec, 0)
    label, fake, prev = synth.make_inputs(spec, 1, size, size, 3100)
    with torch.no_grad():
        oi, om = R.RefGenerator(spec, sd)(label, None, fake, prev)
        pols = [
            ("everything rounded (the bf16 mode)", {}),
            ("filters fp32", {"w": False, "gb_w": False}),
            ("gamma/beta filters fp32 only", {"gb_w": False}),
            ("stored activations fp32 (filters rounded)", {"act": False, "pro": False}),
            ("no second rounding in the prologue", {"pro": False}),
            ("embedder (cond maps) fp32", {"embed": False}),
            ("trunk fp32, mask net + embedder bf16", {"trunk": False}),
            ("mask net fp32, rest bf16", {"mask": False}),
            ("embedder + trunk fp32", {"embed": False, "trunk": False}),
            ("fp16 everywhere (for scale)", {"fmt": "fp16"}),
        ]
        print("%dx%d, seed-0 weights; |error| vs the fp32 oracle" % (size, size))
        print("%-48s %10s %10s %10s %10s" % ("policy", "img max", "img mean", "mask max", "mask mean"))
        for name, pol in pols:
            i, m = Sim(spec, sd, pol).forward(label, fake, prev)
            print("%-48s %10.2e %10.2e %10.2e %10.2e" % (name, float((i - oi).abs().max()), float((i - oi).abs().mean()),
                                                          float((m - om).abs().max()), float((m - om).abs().mean())))


if __name__ == "__main__":
    main()
