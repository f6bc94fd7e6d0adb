"""Config surface of the pose-guided generator (reference: configs/HSM.yaml,
utils/utils.py:77-79 ``get_config``, models/generator.py:46-65,317-324,431-440).

The reference reads a yaml file into an attribute-style dict and every
constructor pulls its hyper-parameters with ``getattr(cfg, key, default)``.
``GenSpec`` resolves exactly those keys with exactly those defaults into a
frozen description of the network the hot path supports, and rejects loudly
every variant the MI355X path does not implement.
"""
from __future__ import annotations

import math
from dataclasses import dataclass

import yaml


class AttrDict(dict):
    """Attribute-style dict (the role easydict.EasyDict plays in the
    reference, utils/utils.py:77-79).  Nested dicts are wrapped."""

    def __init__(self, d=None, **kw):
        super().__init__()
        for k, v in dict(d or {}, **kw).items():
            self[k] = v

    def __setitem__(self, k, v):
        if isinstance(v, dict) and not isinstance(v, AttrDict):
            v = AttrDict(v)
        super().__setitem__(k, v)

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)

    def __setattr__(self, k, v):
        self[k] = v


def get_config(path):
    """yaml file -> attribute dict (reference utils/utils.py:77-79)."""
    with open(path, "r") as stream:
        return AttrDict(yaml.load(stream, Loader=yaml.FullLoader))


def _get(cfg, key, default):
    if cfg is None:
        return default
    if isinstance(cfg, dict):
        return cfg.get(key, default)
    return getattr(cfg, key, default)


# The `gen:` block of configs/HSM.yaml:35-67, the configuration every
# BASELINE.json workload is quoted on.
HSM_GEN = {
    "num_frames_G": 2, "input_image_nc": 3, "input_label_nc": 22,
    "num_filters": 16, "max_num_filters": 512, "num_layers": 6,
    "num_downsamples": 4, "kernel_size": 3,
    "activation_norm_type": "spatially_adaptive",
    "activation_norm_params": {"activation_norm_type": "instance",
                               "num_filters": 0, "kernel_size": 1},
    "weight_norm_type": "spectral", "do_checkpoint": True,
    "mask": {"generate_raw_output": False, "num_filters": 32,
             "max_num_filters": 512, "num_downsamples": 3, "num_res_blocks": 4,
             "kernel_size": 3, "activation_norm_type": "instance",
             "weight_norm_type": "spectral"},
    "embed": {"use_embed": True, "arch": "encoder", "num_filters": 64,
              "max_num_filters": 512, "num_downsamples": 4, "kernel_size": 3,
              "weight_norm_type": "spectral"},
}


def hsm_gen_config(**overrides):
    cfg = AttrDict(HSM_GEN)
    for k, v in overrides.items():
        if isinstance(v, dict) and k in cfg:
            cfg[k].update(v)
        else:
            cfg[k] = v
    return cfg


@dataclass(frozen=True)
class GenSpec:
    """Resolved hyper-parameters of Generator / LabelEmbedder / MaskGenerator."""
    label_nc: int          # gen.input_label_nc            (generator.py:59)
    image_nc: int          # gen.input_image_nc            (generator.py:60)
    num_filters: int       # gen.num_filters               (generator.py:52)
    max_num_filters: int   # gen.max_num_filters           (generator.py:53)
    num_layers: int        # gen.num_layers                (generator.py:48)
    num_down_img: int      # gen.num_downsamples_img, default 4 (generator.py:50)
    num_res_blocks: int    # ceil((num_layers-num_down_img)/2)*2 (generator.py:142-143)
    emb_filters: int       # embed.num_filters             (generator.py:317)
    emb_max_filters: int   # embed.max_num_filters         (generator.py:318)
    emb_down: int          # embed.num_downsamples         (generator.py:65,320)
    mask_filters: int      # mask.num_filters              (generator.py:431)
    mask_max_filters: int  # mask.max_num_filters          (generator.py:432)
    mask_down: int         # mask.num_downsamples          (generator.py:433)
    mask_res_blocks: int   # mask.num_res_blocks           (generator.py:436)

    # channel schedules ---------------------------------------------------
    def nf(self, i):
        """get_num_filters (generator.py:23-32)."""
        return min(self.max_num_filters, self.num_filters * 2 ** i)

    def mask_nf(self, i):
        return min(self.mask_max_filters, self.mask_filters * 2 ** i)

    def emb_ch(self, i):
        return min(self.emb_max_filters, self.emb_filters * 2 ** i)

    def cond_ch(self, i):
        """get_cond_dims (generator.py:269-285); note it clamps with the
        *generator's* max_num_filters, not the embedder's."""
        return min(self.max_num_filters, self.emb_filters * 2 ** min(i, self.emb_down))

    @staticmethod
    def from_cfg(gen_cfg) -> "GenSpec":
        g = gen_cfg
        unsupported = []

        def need(cond, what):
            if not cond:
                unsupported.append(what)

        ks = _get(g, "kernel_size", 3)
        need(ks == 3, "gen.kernel_size=%r (only 3)" % (ks,))
        need(_get(g, "activation_norm_type", None) == "spatially_adaptive",
             "gen.activation_norm_type=%r (only 'spatially_adaptive')"
             % (_get(g, "activation_norm_type", None),))
        anp = _get(g, "activation_norm_params", None)
        need(_get(anp, "activation_norm_type", "sync_batch") == "instance",
             "gen.activation_norm_params.activation_norm_type (only 'instance')")
        need(_get(anp, "num_filters", 0) == 0,
             "gen.activation_norm_params.num_filters > 0 (hidden SPADE MLP)")
        need(_get(anp, "kernel_size", 3) == 1,
             "gen.activation_norm_params.kernel_size=%r (only 1)"
             % (_get(anp, "kernel_size", 3),))
        need(not _get(anp, "partial", False), "partial convolution SPADE")
        need(not _get(anp, "separate_projection", False), "separate_projection SPADE")
        need(_get(g, "weight_norm_type", "spectral") in ("spectral", "none", ""),
             "gen.weight_norm_type (only 'spectral'/'none')")
        emb = _get(g, "embed", None)
        need(emb is not None and bool(_get(emb, "use_embed", "True")),
             "embed.use_embed false")
        need(_get(emb, "arch", "encoderdecoder") == "encoder",
             "embed.arch=%r (only 'encoder')" % (_get(emb, "arch", "encoderdecoder"),))
        need(_get(emb, "kernel_size", 3) == 3, "embed.kernel_size (only 3)")
        need(_get(emb, "activation_norm_type", "none") in ("none", ""),
             "embed.activation_norm_type (only 'none')")
        mask = _get(g, "mask", None)
        need(mask is not None, "gen.mask missing")
        need(_get(mask, "kernel_size", 3) == 3, "mask.kernel_size (only 3)")
        need(_get(mask, "activation_norm_type", "sync_batch") == "instance",
             "mask.activation_norm_type (only 'instance')")
        if unsupported:
            raise NotImplementedError(
                "generator variant not supported by the MI355X path: "
                + "; ".join(unsupported))

        num_layers = _get(g, "num_layers", 7)
        num_down_img = _get(g, "num_downsamples_img", 4)
        spec = GenSpec(
            label_nc=g["input_label_nc"] if isinstance(g, dict) else g.input_label_nc,
            image_nc=g["input_image_nc"] if isinstance(g, dict) else g.input_image_nc,
            num_filters=_get(g, "num_filters", 32),
            max_num_filters=_get(g, "max_num_filters", 1024),
            num_layers=num_layers,
            num_down_img=num_down_img,
            num_res_blocks=int(math.ceil((num_layers - num_down_img) / 2.0) * 2),
            emb_filters=_get(emb, "num_filters", 32),
            emb_max_filters=_get(emb, "max_num_filters", 1024),
            emb_down=_get(emb, "num_downsamples", 5),
            mask_filters=_get(mask, "num_filters", 32),
            mask_max_filters=_get(mask, "max_num_filters", 1024),
            mask_down=_get(mask, "num_downsamples", 5),
            mask_res_blocks=_get(mask, "num_res_blocks", 6),
        )
        if spec.emb_down < spec.num_down_img:
            # generator.py:204 indexes cond_maps[min(emb_down, i)]; resolutions
            # only line up when the embedder has at least as many levels.
            raise NotImplementedError("embed.num_downsamples < num_downsamples_img")
        if spec.emb_down != spec.num_down_img:
            raise NotImplementedError(
                "embed.num_downsamples != num_downsamples_img (HSM.yaml uses 4/4)")
        return spec

    @property
    def size_multiple(self):
        """H and W must be multiples of this."""
        return 2 ** max(self.num_down_img, self.mask_down)
