"""TEST INFRASTRUCTURE (checker only; see oracle/__init__.py).

CPU model of the 16-bit storage modes' roundings (round 3's tools/probes/bf16_policy_sim.py, promoted in round 6): the
oracle's forward (oracle/generator_ref.py) re-run with the tensors rounded to bf16 / IEEE half at exactly the places the
16-bit kernels round them:

  * every tensor a kernel stores (conv outputs after bias / residual / activation, SPADE outputs, pooled tensors, joins),
  * the prologue's result on its way into LDS (IN affine + LeakyReLU is fp32 arithmetic, then rounded again),
  * the filters of the matrix-core kernels (gamma/beta filters included); fp32 accumulation = exact products of the
    rounded operands summed in fp32.

`predict()` gives the error of a mode against the fp32 oracle on given inputs: what the FORMAT costs.  The GPU kernels
are expected to sit on it (they add nothing to the format's rounding): tests/test_precision_model.py compares the
committed GPU figures with it, so the 16-bit bounds are derived from a model and not only from what was once measured.
A policy dict switches groups of the roundings off (tools/probes/bf16_policy_sim.py prints the table of DESIGN 6).
"""
from __future__ import annotations

import torch
import torch.nn.functional as F

from . import generator_ref as R


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


@torch.no_grad()
def predict(spec, sd, label, fake, prev, fmt="bf16", pol=None):
    """{max_abs_img, mean_abs_img, max_abs_mask, mean_abs_mask} of the rounded forward against the fp32 oracle."""
    p = dict(pol or {})
    p["fmt"] = "fp16" if fmt in ("f16", "fp16") else "bf16"
    oi, om = R.RefGenerator(spec, sd)(label, None, fake, prev)
    i, m = Sim(spec, sd, p).forward(label, fake, prev)
    return {"max_abs_img": float((i - oi).abs().max()), "mean_abs_img": float((i - oi).abs().mean()),
            "max_abs_mask": float((m - om).abs().max()), "mean_abs_mask": float((m - om).abs().mean())}
