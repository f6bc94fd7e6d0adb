"""TEST INFRASTRUCTURE (checker only; see oracle/__init__.py).

CPU restatement of the reference generator forward in plain PyTorch fp32
functional ops, NCHW.  Every function cites the reference lines it restates;
``PGNR`` = /root/reference/Pose_Guided_Neural_Rendering.

Pinned against the imported reference generator: tests/golden/make_golden.py
(run in the build container) asserts max-abs <= 1e-5 on every committed case
and stores the reference's outputs as fixtures.
"""
from __future__ import annotations

from typing import Dict, List, Optional

import numpy as np
import torch
import torch.nn.functional as F

LRELU_SLOPE = 0.2   # PGNR/models/layers/nonlinearity.py:21-22
IN_EPS = 1e-5       # nn.InstanceNorm2d default (PGNR/models/layers/activation_norm.py:399-402)


# --------------------------------------------------------------------------
# weights
# --------------------------------------------------------------------------
def fold_spectral_norm(w_orig: torch.Tensor, u: torch.Tensor, v: torch.Tensor) -> torch.Tensor:
    """Eval-mode torch.nn.utils.spectral_norm (hook installed at
    PGNR/models/layers/weight_norm.py:84-85): no power iteration,
    ``W = W_orig / (u . (W_mat v))`` with u, v as stored."""
    w_mat = w_orig.reshape(w_orig.shape[0], -1)
    sigma = torch.dot(u, torch.mv(w_mat, v))
    return w_orig / sigma


def conv_weight(sd: Dict[str, torch.Tensor], prefix: str):
    """(weight, bias) of ``<prefix>.layers.conv`` with spectral norm folded."""
    p = prefix + ".layers.conv"
    if p + ".weight_orig" in sd:
        w = fold_spectral_norm(sd[p + ".weight_orig"], sd[p + ".weight_u"], sd[p + ".weight_v"])
    else:
        w = sd[p + ".weight"]
    return w, sd[p + ".bias"]


# --------------------------------------------------------------------------
# layer restatements
# --------------------------------------------------------------------------
def lrelu(x):
    return F.leaky_relu(x, LRELU_SLOPE)


def instance_norm(x, weight=None, bias=None):
    """nn.InstanceNorm2d(eps=1e-5, track_running_stats=False): per (n,c) over
    H*W, biased variance (activation_norm.py:399-402)."""
    mean = x.mean(dim=(2, 3), keepdim=True)
    var = x.var(dim=(2, 3), unbiased=False, keepdim=True)
    y = (x - mean) * torch.rsqrt(var + IN_EPS)
    if weight is not None:
        y = y * weight.view(1, -1, 1, 1) + bias.view(1, -1, 1, 1)
    return y


def spade(sd, prefix, x, cond):
    """SpatiallyAdaptiveNorm.forward (activation_norm.py:211-234) with
    num_filters=0, kernel_size=1: IN(x) * (1 + gamma) + beta, [gamma|beta] =
    chunk(conv1x1(nearest_resize(cond)))."""
    p = prefix + ".layers.norm.mlps.0.0.layers.conv"
    if cond.shape[2:] != x.shape[2:]:
        cond = F.interpolate(cond, size=x.shape[2:], mode="nearest")
    gb = F.conv2d(cond, sd[p + ".weight"], sd[p + ".bias"])
    gamma, beta = gb.chunk(2, dim=1)
    return instance_norm(x) * (1 + gamma) + beta


def spade_res_block(sd, name, x, cond, taps=None):
    """Res2dBlock(order='NACNAC') forward (residual.py:115-151; block order
    strings residual.py:82-108; _BaseConvBlock.forward conv.py:77-91)."""
    w0, b0 = conv_weight(sd, name + ".conv_block_0")
    w1, b1 = conv_weight(sd, name + ".conv_block_1")
    ys0 = lrelu(spade(sd, name + ".conv_block_0", x, cond))
    h = F.conv2d(ys0, w0, b0, padding=1)
    y1 = lrelu(spade(sd, name + ".conv_block_1", h, cond))
    dx = F.conv2d(y1, w1, b1, padding=1)
    if taps is not None:
        taps[name + ".ys0"] = ys0
        taps[name + ".y1"] = y1
    if (name + ".conv_block_s.layers.conv.bias") in sd:
        ws, bs = conv_weight(sd, name + ".conv_block_s")
        xs = F.conv2d(spade(sd, name + ".conv_block_s", x, cond), ws, bs)   # 'NC': no activation
    else:
        xs = x
    if taps is not None:
        taps[name + ".h"] = h
    return xs + dx


def conv_in_lrelu(sd, name, x, stride=1, act=True, taps=None, tapname=None):
    """Conv2dBlock(order='CNA') with InstanceNorm2d(affine=True) (generator.py:442-459)."""
    w, b = conv_weight(sd, name)
    y = F.conv2d(x, w, b, stride=stride, padding=w.shape[-1] // 2)
    if taps is not None and tapname:
        taps[tapname] = y
    y = instance_norm(y, sd[name + ".layers.norm.weight"], sd[name + ".layers.norm.bias"])
    return lrelu(y) if act else y


def mask_res_block(sd, name, x):
    """Res2dBlock(order='CNACN') (generator.py:465-476; residual.py:82-108,129-151)."""
    dx = conv_in_lrelu(sd, name + ".conv_block_0", x)
    dx = conv_in_lrelu(sd, name + ".conv_block_1", dx, act=False)
    if (name + ".conv_block_s.layers.conv.bias") in sd:
        xs = conv_in_lrelu(sd, name + ".conv_block_s", x, act=False)
    else:
        xs = x
    return xs + dx


class RefGenerator:
    """Restatement of Generator (PGNR/models/generator.py:35-302)."""

    def __init__(self, spec, state_dict, dtype=torch.float32):
        """dtype=torch.float64: the same restatement evaluated in double precision (inputs are converted on entry) - the
        yardstick fp32 implementations are ranked against (tools/products_error.py, tests/test_precision_model.py)."""
        self.spec = spec
        self.dtype = dtype
        self.sd = {k: v.detach().to(dtype) for k, v in state_dict.items()}

    # LabelEmbedder.forward, arch 'encoder' (generator.py:360-387)
    def embed(self, x):
        w, b = conv_weight(self.sd, "ref_embedding.conv_first")
        out = [lrelu(F.conv2d(x, w, b, padding=1))]
        for i in range(self.spec.emb_down):
            w, b = conv_weight(self.sd, "ref_embedding.down_%d" % i)
            out.append(lrelu(F.conv2d(out[-1], w, b, stride=2, padding=1)))
        return out

    # MaskGenerator.forward (generator.py:493-510)
    def mask_net(self, label, img9, taps=None):
        sd, m = self.sd, "flow_network_temp"
        a, b = label, img9
        for i in range(self.spec.mask_down + 1):
            a = conv_in_lrelu(sd, "%s.down_lbl.%d" % (m, i), a, stride=1 if i == 0 else 2,
                              taps=taps, tapname="mask.lbl_%d.raw" % i)
            b = conv_in_lrelu(sd, "%s.down_img.%d" % (m, i), b, stride=1 if i == 0 else 2,
                              taps=taps, tapname="mask.img_%d.raw" % i)
        r = torch.cat([a, b], dim=1)
        if taps is not None:
            taps["mask.cat"] = r
            taps["mask.cat.raw"] = torch.cat([taps.pop("mask.lbl_%d.raw" % self.spec.mask_down),
                                              taps.pop("mask.img_%d.raw" % self.spec.mask_down)], dim=1)
        for i in range(self.spec.mask_res_blocks):
            r = mask_res_block(sd, "%s.res_flow.%d" % (m, i), r)
            if taps is not None:
                taps["mask.res_%d" % i] = r
        for j in range(self.spec.mask_down):
            r = F.interpolate(r, scale_factor=2, mode="nearest")       # nn.Upsample(scale_factor=2)
            r = conv_in_lrelu(sd, "%s.up_flow.%d" % (m, 2 * j + 1), r, taps=taps,
                              tapname="mask.up_%d.raw" % j)
            if taps is not None:
                taps["mask.up_%d" % j] = r
        w, bb = conv_weight(sd, m + ".conv_mask.0")
        return torch.sigmoid(F.conv2d(r, w, bb, padding=1))

    # Generator.forward (generator.py:181-234); label_prev is dead
    @torch.no_grad()
    def forward(self, label, label_prev, img_fake, img_prev, taps: Optional[dict] = None):
        sp, sd = self.spec, self.sd
        label, img_fake, img_prev = label.to(self.dtype), img_fake.to(self.dtype), img_prev.to(self.dtype)
        cond = self.embed(torch.cat([img_fake, img_prev], dim=1))            # :197
        if taps is not None:
            for i, c in enumerate(cond):
                taps["cond_%d" % i] = c
        w, b = conv_weight(sd, "down_first")
        x = F.conv2d(label, w, b, padding=1)                                  # :201
        if taps is not None:
            taps["down_first"] = x
        for i in range(sp.num_down_img + 1):                                  # :203-208
            x = spade_res_block(sd, "down_%d" % i, x, cond[min(sp.emb_down, i)], taps)
            if taps is not None:
                taps["down_%d" % i] = x
            if i != sp.num_down_img:
                x = F.avg_pool2d(x, 3, stride=2, padding=1)                   # :127 count_include_pad=True
        j = min(sp.emb_down, sp.num_down_img + 1)
        for i in range(sp.num_res_blocks):                                    # :211-215
            x = spade_res_block(sd, "res_%d" % i, x, cond[j], taps)
            if taps is not None:
                taps["res_%d" % i] = x
        for i in range(sp.num_down_img, -1, -1):                              # :220-224, :236-250
            x = spade_res_block(sd, "up_%d" % i, x, cond[min(i, sp.emb_down)], taps)
            if taps is not None:
                taps["up_%d" % i] = x
            if i != 0:
                x = F.interpolate(x, scale_factor=2, mode="nearest")
        w, b = conv_weight(sd, "conv_img")
        img = torch.tanh(F.conv2d(lrelu(x), w, b, padding=1))                 # :114-116 'AC', :228
        mask = self.mask_net(label, torch.cat([img_prev, img_fake, img], dim=1), taps)  # :232
        return img, mask

    __call__ = forward


# --------------------------------------------------------------------------
# driver restatement (PGNR/models/evaluator.py:238-266)
# --------------------------------------------------------------------------
def blend(img, mask, dain):
    """evaluator.py:256-258."""
    m3 = mask.repeat(1, 3, 1, 1)
    return img * m3 + dain * (1 - m3)


@torch.no_grad()
def autoregressive_segment(gen, key_frame, labels: List[torch.Tensor], dains: List[torch.Tensor]):
    """One segment between key-frames: prev starts as the ground-truth key
    frame (evaluator.py:240-244) and every following frame feeds on the
    previous *fused* frame (evaluator.py:252-262).  Returns lists
    (img, mask, fuse) for the generated frames."""
    prev = key_frame
    imgs, masks, fuses = [], [], []
    for lab, dain in zip(labels, dains):
        img, mask = gen(lab, None, dain, prev)
        fuse = blend(img, mask, dain)
        imgs.append(img); masks.append(mask); fuses.append(fuse)
        prev = fuse
    return imgs, masks, fuses


def quantise_uint8(image_tensor: torch.Tensor) -> np.ndarray:
    """tensor2images for a 3-channel image (utils/utils.py:122-147):
    HWC, x*0.5+0.5, clip to [0,1], *255, astype(uint8) (C truncation).
    The reference does this arithmetic in float64 (numpy mean/std arrays)."""
    x = image_tensor[0].cpu().float().numpy()
    x = np.transpose(x, (1, 2, 0)) * np.array([0.5, 0.5, 0.5]) + np.array([0.5, 0.5, 0.5])
    x = np.clip(x, 0, 1) * 255.0
    return x.astype(np.uint8)


def sample_rate_of(num_pose: int, num_key: int) -> int:
    """evaluator.py:190."""
    return 2 ** int(np.log2((num_pose - 1) / (num_key - 1)))
