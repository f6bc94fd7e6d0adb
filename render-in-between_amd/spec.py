"""Layer inventory of the generator and the checkpoint (state-dict) layout.

This is the build's own statement of which convolutions exist, in what order
they run and what their checkpoint tensors are called; it is derived from
reading the reference constructors (models/generator.py:43-178,315-358,
423-491; layers/conv.py:14-91; layers/residual.py:20-113;
layers/activation_norm.py:134-209) and is pinned against the reference's real
``state_dict()`` by tests/golden/state_dict_keys.json.

Checkpoint naming rules reproduced here:
  * a conv wrapped in spectral norm stores ``weight_orig, weight_u, weight_v,
    bias``; a plain conv stores ``weight, bias``
    (torch.nn.utils.spectral_norm via layers/weight_norm.py:84-85);
  * a SPADE norm stores its gamma/beta 1x1 conv as
    ``<block>.layers.norm.mlps.0.0.layers.conv.{weight,bias}`` (no spectral
    norm: SpatiallyAdaptiveNorm passes weight_norm_type='');
  * the mask network's InstanceNorm2d(affine=True) stores
    ``<block>.layers.norm.{weight,bias}``.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import List, Optional, Tuple

from .config import GenSpec


@dataclass
class ConvSpec:
    """One convolution of the path (a `C` of a Conv2dBlock order string)."""
    name: str                 # state-dict prefix, e.g. 'down_0.conv_block_0'
    cin: int
    cout: int
    ksize: int                # 3 or 1
    stride: int = 1
    spectral: bool = True
    # SPADE in front of the conv ('NAC' / 'NC' orders): cond channels, else 0
    spade_cond: int = 0
    # InstanceNorm2d(affine=True) after the conv ('CNA' / 'CN' orders)
    in_affine: bool = False
    used: bool = True         # False: lives in the checkpoint, never executed

    @property
    def conv_prefix(self):
        return self.name + ".layers.conv"

    @property
    def spade_prefix(self):
        return self.name + ".layers.norm.mlps.0.0.layers.conv"

    @property
    def norm_prefix(self):
        return self.name + ".layers.norm"

    def tensors(self) -> List[Tuple[str, Tuple[int, ...]]]:
        out = []
        if self.spade_cond:
            out.append((self.spade_prefix + ".weight", (2 * self.cin, self.spade_cond, 1, 1)))
            out.append((self.spade_prefix + ".bias", (2 * self.cin,)))
        k = self.ksize
        if self.spectral:
            out.append((self.conv_prefix + ".bias", (self.cout,)))
            out.append((self.conv_prefix + ".weight_orig", (self.cout, self.cin, k, k)))
            out.append((self.conv_prefix + ".weight_u", (self.cout,)))
            out.append((self.conv_prefix + ".weight_v", (self.cin * k * k,)))
        else:
            out.append((self.conv_prefix + ".weight", (self.cout, self.cin, k, k)))
            out.append((self.conv_prefix + ".bias", (self.cout,)))
        if self.in_affine:
            out.append((self.norm_prefix + ".weight", (self.cout,)))
            out.append((self.norm_prefix + ".bias", (self.cout,)))
        return out


def _embedder(prefix: str, spec: GenSpec, cin: int, used: bool) -> List[ConvSpec]:
    """LabelEmbedder, arch 'encoder' (generator.py:315-348)."""
    convs = [ConvSpec(prefix + ".conv_first", cin, spec.emb_ch(0), 3, used=used)]
    for i in range(spec.emb_down):
        convs.append(ConvSpec(prefix + ".down_%d" % i, spec.emb_ch(i), spec.emb_ch(i + 1),
                              3, stride=2, used=used))
    return convs


def _spade_res_block(name: str, cin: int, cout: int, cond: int) -> List[ConvSpec]:
    """Res2dBlock(order='NACNAC') (residual.py:20-113)."""
    hidden = min(cin, cout)
    convs = [ConvSpec(name + ".conv_block_0", cin, hidden, 3, spade_cond=cond),
             ConvSpec(name + ".conv_block_1", hidden, cout, 3, spade_cond=cond)]
    if cin != cout:
        convs.append(ConvSpec(name + ".conv_block_s", cin, cout, 1, spade_cond=cond))
    return convs


def _mask_res_block(name: str, cin: int, cout: int) -> List[ConvSpec]:
    """Res2dBlock(order='CNACN') with affine instance norm (generator.py:465-476)."""
    hidden = min(cin, cout)
    convs = [ConvSpec(name + ".conv_block_0", cin, hidden, 3, in_affine=True),
             ConvSpec(name + ".conv_block_1", hidden, cout, 3, in_affine=True)]
    if cin != cout:
        convs.append(ConvSpec(name + ".conv_block_s", cin, cout, 1, in_affine=True))
    return convs


def conv_inventory(spec: GenSpec) -> List[ConvSpec]:
    """Every convolution that owns checkpoint tensors, in the order the
    reference registers them (generator.py:66-68,104-120,146-178)."""
    L: List[ConvSpec] = []
    L += _embedder("ref_embedding", spec, spec.image_nc * 2, used=True)
    L += _embedder("label_embedding", spec, spec.label_nc, used=False)
    for i in range(spec.num_down_img, -1, -1):
        L += _spade_res_block("up_%d" % i, spec.nf(i + 1), spec.nf(i), spec.cond_ch(i))
    L.append(ConvSpec("conv_img", spec.num_filters, spec.image_nc, 3, spectral=False))
    L.append(ConvSpec("conv_mask", spec.num_filters, 1, 3, spectral=False, used=False))
    L.append(ConvSpec("down_first", spec.label_nc, spec.num_filters, 3, spectral=False))
    for i in range(spec.num_down_img + 1):
        L += _spade_res_block("down_%d" % i, spec.nf(i), spec.nf(i + 1), spec.cond_ch(i))
    res_ch = spec.nf(spec.num_down_img + 1)
    for i in range(spec.num_res_blocks):
        L += _spade_res_block("res_%d" % i, res_ch, res_ch, spec.cond_ch(spec.num_down_img + 1))
    # MaskGenerator (generator.py:423-491)
    m = "flow_network_temp"
    for branch, cin in (("down_lbl", spec.label_nc), ("down_img", spec.image_nc * 3)):
        L.append(ConvSpec("%s.%s.0" % (m, branch), cin, spec.mask_filters, 3, in_affine=True))
        for i in range(spec.mask_down):
            L.append(ConvSpec("%s.%s.%d" % (m, branch, i + 1), spec.mask_nf(i),
                              spec.mask_nf(i + 1), 3, stride=2, in_affine=True))
    ch = spec.mask_nf(spec.mask_down)
    for i in range(spec.mask_res_blocks):
        L += _mask_res_block("%s.res_flow.%d" % (m, i), ch * 2 if i == 0 else ch, ch)
    for j, i in enumerate(reversed(range(spec.mask_down))):
        # nn.Sequential of [Upsample, conv] pairs: the conv sits at odd indices
        L.append(ConvSpec("%s.up_flow.%d" % (m, 2 * j + 1), spec.mask_nf(i + 1),
                          spec.mask_nf(i), 3, in_affine=True))
    L.append(ConvSpec(m + ".conv_mask.0", spec.mask_filters, 1, 3, spectral=False))
    return L


def state_dict_spec(spec: GenSpec) -> List[Tuple[str, Tuple[int, ...]]]:
    """(name, shape) of every tensor of the reference Generator.state_dict()."""
    out = []
    for c in conv_inventory(spec):
        out += c.tensors()
    return out


def conv_flops(spec: GenSpec, H: int, W: int) -> float:
    """2*MAC over the executed convolutions (231.60 GFLOP @512^2),
    counting each conv at its output resolution."""
    inv = {c.name: c for c in conv_inventory(spec) if c.used}
    total = 0.0

    def add(c: ConvSpec, ho, wo):
        nonlocal total
        total += 2.0 * c.cin * c.ksize * c.ksize * c.cout * ho * wo
        if c.spade_cond:
            # gamma/beta 1x1 conv runs at the conv's *input* resolution == ho,wo
            # (all SPADE convs on the path are stride 1)
            total += 2.0 * c.spade_cond * 2 * c.cin * ho * wo

    h, w = H, W
    add(inv["ref_embedding.conv_first"], h, w)
    for i in range(spec.emb_down):
        h, w = h // 2, w // 2
        add(inv["ref_embedding.down_%d" % i], h, w)
    add(inv["down_first"], H, W)
    add(inv["conv_img"], H, W)
    for i in range(spec.num_down_img + 1):
        s = 2 ** i
        for blk in ("0", "1", "s"):
            add(inv["down_%d.conv_block_%s" % (i, blk)], H // s, W // s)
            add(inv["up_%d.conv_block_%s" % (i, blk)], H // s, W // s)
    s = 2 ** spec.num_down_img
    for i in range(spec.num_res_blocks):
        for blk in ("0", "1"):
            add(inv["res_%d.conv_block_%s" % (i, blk)], H // s, W // s)
    m = "flow_network_temp"
    for branch in ("down_lbl", "down_img"):
        for i in range(spec.mask_down + 1):
            s = 2 ** i
            add(inv["%s.%s.%d" % (m, branch, i)], H // s, W // s)
    s = 2 ** spec.mask_down
    for i in range(spec.mask_res_blocks):
        for blk in ("0", "1", "s"):
            k = "%s.res_flow.%d.conv_block_%s" % (m, i, blk)
            if k in inv:
                add(inv[k], H // s, W // s)
    for j, i in enumerate(reversed(range(spec.mask_down))):
        s = 2 ** i
        add(inv["%s.up_flow.%d" % (m, 2 * j + 1)], H // s, W // s)
    add(inv[m + ".conv_mask.0"], H, W)
    return total
