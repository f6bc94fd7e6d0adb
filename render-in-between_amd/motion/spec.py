"""Configuration and checkpoint layout of the motion transformer.

The layout restates what the reference's constructors create (HMM/models/transformer.py:20-46,
184-200, 257-276: nn.Linear / nn.MultiheadAttention / nn.LayerNorm members) and is pinned against
the reference's real ``state_dict()`` keys by tests/golden/motion_state_dict_keys.json.
"""
from __future__ import annotations

from dataclasses import dataclass, asdict
from typing import List, Tuple

ACTIVATIONS = ("relu", "gelu", "leaky_relu")   # HMM/models/transformer.py:365-375 ('glu' changes the width: rejected)


@dataclass(frozen=True)
class MotionSpec:
    """``transformer:`` / ``pos_encode:`` sections of HMM/configs/config.yaml."""
    input_joints: int = 38
    hidden_dim: int = 128
    nheads: int = 8
    dim_feedforward: int = 256
    enc_layers: int = 6
    dec_layers: int = 6
    activation: str = "leaky_relu"
    pre_norm: bool = True
    two_stage: bool = True
    pos_hidden_dim: int = 128

    @staticmethod
    def from_cfg(cfg) -> "MotionSpec":
        """cfg: the whole yaml (mapping or attribute dict) or its ``transformer`` section."""
        get = (lambda o, k, d=None: o.get(k, d)) if isinstance(cfg, dict) else (lambda o, k, d=None: getattr(o, k, d))
        t = get(cfg, "transformer", None) or cfg
        pe = get(cfg, "pos_encode", None)
        tg = (lambda k, d: t.get(k, d)) if isinstance(t, dict) else (lambda k, d: getattr(t, k, d))
        if pe is not None:
            pg = (lambda k, d: pe.get(k, d)) if isinstance(pe, dict) else (lambda k, d: getattr(pe, k, d))
            kind = pg("position_embedding", "v2")
            if kind not in ("v2", "sine"):
                # 'v3' / 'learned' (HMM/models/position_encoding.py:55-90) is an nn.Embedding(160, hidden_dim) that the reference
                # creates in Trainer.__init__ (models/trainer.py:64) and never restores: load_state_dict is called on the
                # transformer alone (:74) and inference hands over trainer.pos_encode as constructed (inference.py:67).  At
                # inference its table is therefore fresh uniform noise, different in every process: there is no defined result
                # to reproduce, so the variant is refused instead of built.
                raise ValueError("position_embedding '%s' is not supported: only the sine encoding 'v2' is defined at inference (the reference "
                                 "never saves or loads the learned table, so its inference would add fresh random noise to every token)" % kind)
            pos_dim = int(pg("hidden_dim", tg("hidden_dim", 128)))
        else:
            pos_dim = int(tg("hidden_dim", 128))
        if tg("intermediate", False):
            # return_intermediate_dec (HMM/models/transformer.py:162-196) makes the decoder return the STACK of its layers' outputs
            # [layers, L, N, C] for the training losses; the reference's own inference then fails on it (inference.py:
            # `pred.permute(1, 2, 0)` of a 4-D tensor), so the inference path this package replaces never runs with it
            raise ValueError("transformer.intermediate=True is a training-time option (the decoder returns the stack of its layers' outputs, "
                             "which the reference's own inference cannot consume either); set it to False for inference")
        s = MotionSpec(input_joints=int(tg("input_joints", 38)), hidden_dim=int(tg("hidden_dim", 128)),
                       nheads=int(tg("nheads", 8)), dim_feedforward=int(tg("dim_feedforward", 256)),
                       enc_layers=int(tg("enc_layers", 6)), dec_layers=int(tg("dec_layers", 6)),
                       activation=str(tg("activation", "leaky_relu")), pre_norm=bool(tg("pre_norm", True)),
                       two_stage=bool(tg("two_stage", True)), pos_hidden_dim=pos_dim)
        s.validate()
        return s

    def validate(self):
        if self.activation not in ACTIVATIONS:
            raise ValueError("activation should be relu/gelu/leaky_relu, not %s" % self.activation)
        if self.hidden_dim % self.nheads != 0:
            raise ValueError("hidden_dim must be divisible by nheads")
        if self.pos_hidden_dim != self.hidden_dim:
            raise ValueError("pos_encode.hidden_dim must equal transformer.hidden_dim (the encoding is added to the tokens)")
        if self.hidden_dim % 4 != 0 or self.hidden_dim > 256 or self.hidden_dim // self.nheads > 64:
            raise ValueError("unsupported width: hidden_dim <= 256 (multiple of 4), head_dim <= 64")

    def as_dict(self):
        return asdict(self)


def state_dict_spec(spec: MotionSpec) -> List[Tuple[str, Tuple[int, ...]]]:
    """(name, shape) of every checkpoint tensor, in the reference's ``state_dict()`` order."""
    D, F, C = spec.hidden_dim, spec.dim_feedforward, spec.input_joints
    out: List[Tuple[str, Tuple[int, ...]]] = [("input_embed.weight", (D, C)), ("input_embed.bias", (D,))]

    def attn(p):
        return [(p + ".in_proj_weight", (3 * D, D)), (p + ".in_proj_bias", (3 * D,)),
                (p + ".out_proj.weight", (D, D)), (p + ".out_proj.bias", (D,))]

    def ffn(p):
        return [(p + ".linear1.weight", (F, D)), (p + ".linear1.bias", (F,)),
                (p + ".linear2.weight", (D, F)), (p + ".linear2.bias", (D,))]

    def norm(p):
        return [(p + ".weight", (D,)), (p + ".bias", (D,))]
    for i in range(spec.enc_layers):
        p = "encoder.layers.%d" % i
        out += attn(p + ".self_attn") + ffn(p) + norm(p + ".norm1") + norm(p + ".norm2")
    if spec.pre_norm:
        out += norm("encoder.norm")
    for i in range(spec.dec_layers):
        p = "decoder.layers.%d" % i
        out += attn(p + ".self_attn") + attn(p + ".multihead_attn") + ffn(p)
        out += norm(p + ".norm1") + norm(p + ".norm2") + norm(p + ".norm3")
    out += norm("decoder.norm")
    out += [("joints_embed.weight", (C, D)), ("joints_embed.bias", (C,))]
    return out
