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
import torch.nn.functional as F

import render_in_between_amd as rib
from render_in_between_amd import synth
from oracle import generator_ref as R


def rb(x):
    return x.to(torch.bfloat16).to(torch.float32)


def rh(x):
    return x.to(torch.float16).to(torch.float32)


class Sim:
    """pol: dict of booleans
         w      round conv filters            gb_w   round gamma/beta filters
         act    round stored activations      pro    round the prologue result again (mask network)
         cond   round the condition maps      trunk / mask / embed: apply `act` + `w` inside that sub-network
         fmt    'bf16' | 'fp16'
    """

    def __init__(self, spec, sd, pol):
        self.spec, self.sd, self.pol = spec, {k: v.float() for k, v in sd.items()}, pol
        self.r = rh if pol.get("fmt") == "fp16" else rb

    def qa(self, x, net):
        return self.r(x) if self.pol.get("act", True) and self.pol.get(net, True) else x

    def qw(self, w, net, gb=False):
        on = self.pol.get("gb_w" if gb else "w", True) and self.pol.get(net, True)
        return self.r(w) if on else w

    def conv(self, x, name, net, stride=1, padding=1):
        w, b = R.conv_weight(self.sd, name)
        return F.conv2d(x, self.qw(w, net), b, stride=stride, padding=padding)

    def spade(self, prefix, x, cond, net="trunk"):
        p = prefix + ".layers.norm.mlps.0.0.layers.conv"
        gb = F.conv2d(cond, self.qw(self.sd[p + ".weight"], net, gb=True), self.sd[p + ".bias"])
        g, b = gb.chunk(2, dim=1)
        return R.instance_norm(x) * (1 + g) + b

    def block(self, name, x, cond):
        ys0 = self.qa(R.lrelu(self.spade(name + ".conv_block_0", x, cond)), "trunk")
        h = self.qa(self.conv(ys0, name + ".conv_block_0", "trunk"), "trunk")
        y1 = self.qa(R.lrelu(self.spade(name + ".conv_block_1", h, cond)), "trunk")
        dx = self.conv(y1, name + ".conv_block_1", "trunk")
        if (name + ".conv_block_s.layers.conv.bias") in self.sd:
            ys1 = self.qa(self.spade(name + ".conv_block_s", x, cond), "trunk")
            xs = self.conv(ys1, name + ".conv_block_s", "trunk", padding=0)
        else:
            xs = x
        return self.qa(xs + dx, "trunk")

    def cna(self, name, x, stride=1, act=True, first=False):
        # producer stores the raw conv output (rounded); the consumer applies IN affine + lrelu in fp32 and rounds again
        y = self.qa(self.conv(x, name, "mask", stride=stride, padding=R.conv_weight(self.sd, name)[0].shape[-1] // 2), "mask")
        y = R.instance_norm(y, self.sd[name + ".layers.norm.weight"], self.sd[name + ".layers.norm.bias"])
        y = R.lrelu(y) if act else y
        return self.r(y) if (self.pol.get("pro", True) and self.pol.get("mask", True)) else y

    def forward(self, label, fake, prev):
        sp = self.spec
        x = torch.cat([fake, prev], dim=1)
        e = "embed"
        cond = [self.qa(R.lrelu(self.conv(x, "ref_embedding.conv_first", e)), e)]
        for i in range(sp.emb_down):
            cond.append(self.qa(R.lrelu(self.conv(cond[-1], "ref_embedding.down_%d" % i, e, stride=2)), e))
        if not self.pol.get("cond", True):
            pass
        x = self.qa(self.conv(label, "down_first", "trunk"), "trunk")
        for i in range(sp.num_down_img + 1):
            x = self.block("down_%d" % i, x, cond[min(sp.emb_down, i)])
            if i != sp.num_down_img:
                x = self.qa(F.avg_pool2d(x, 3, stride=2, padding=1), "trunk")
        j = min(sp.emb_down, sp.num_down_img + 1)
        for i in range(sp.num_res_blocks):
            x = self.block("res_%d" % i, x, cond[j])
        for i in range(sp.num_down_img, -1, -1):
            x = self.block("up_%d" % i, x, cond[min(i, sp.emb_down)])
            if i != 0:
                x = F.interpolate(x, scale_factor=2, mode="nearest")
        w, b = R.conv_weight(self.sd, "conv_img")
        img = torch.tanh(F.conv2d(R.lrelu(x), w, b, padding=1))          # head: fp32 filters
        m = "flow_network_temp"
        a, bb = label, torch.cat([prev, fake, img], dim=1)
        for i in range(sp.mask_down + 1):
            a = self.cna("%s.down_lbl.%d" % (m, i), a, stride=1 if i == 0 else 2)
            bb = self.cna("%s.down_img.%d" % (m, i), bb, stride=1 if i == 0 else 2)
        r = torch.cat([a, bb], dim=1)
        for i in range(sp.mask_res_blocks):
            n = "%s.res_flow.%d" % (m, i)
            dx = self.cna(n + ".conv_block_0", r)
            dx = self.cna(n + ".conv_block_1", dx, act=False)
            xs = self.cna(n + ".conv_block_s", r, act=False) if (n + ".conv_block_s.layers.conv.bias") in self.sd else r
            r = self.qa(xs + dx, "mask")
        for jn in range(sp.mask_down):
            r = F.interpolate(r, scale_factor=2, mode="nearest")
            r = self.cna("%s.up_flow.%d" % (m, 2 * jn + 1), r)
        w, b2 = R.conv_weight(self.sd, m + ".conv_mask.0")
        return img, torch.sigmoid(F.conv2d(r, w, b2, padding=1))


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
