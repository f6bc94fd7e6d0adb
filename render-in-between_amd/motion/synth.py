"""Seed-defined synthetic checkpoint and inputs for the motion transformer (the published
``model_epoch399.pth`` is not in the tree): Xavier-uniform matrices as the reference initialises them
(HMM/models/transformer.py:48-51), non-trivial biases and LayerNorm affines so that every term of
the forward is exercised.  A pure function of the seed: runs identically on the build box (golden
generation) and the GPU box (tests, bench)."""
from __future__ import annotations

import numpy as np
import torch

from .spec import MotionSpec, state_dict_spec


def make_state_dict(spec: MotionSpec, seed: int = 0):
    sd = {}
    for idx, (name, shape) in enumerate(state_dict_spec(spec)):
        rng = np.random.default_rng([int(seed), 7001, idx])
        if len(shape) == 2:
            bound = float(np.sqrt(6.0 / (shape[0] + shape[1])))
            a = rng.uniform(-bound, bound, shape)
        elif ".norm" in name and name.endswith(".weight"):
            a = 1.0 + 0.1 * rng.standard_normal(shape)
        else:
            a = 0.05 * rng.standard_normal(shape)
        sd[name] = torch.from_numpy(a.astype(np.float32))
    return sd


def make_clip(spec: MotionSpec, n_key: int, rate: int, seed: int = 0):
    """A smooth random clip in network coordinates: (input [C][L] with the non-key frames zeroed,
    interp [C][L], encoder_mask bool [L] (True = not a key frame), decoder_mask bool [L] (all False))."""
    L = (n_key - 1) * rate + 1
    rng = np.random.default_rng([int(seed), 7002])
    t = np.linspace(0.0, 1.0, L)[None, :]
    C = spec.input_joints
    x = sum(rng.standard_normal((C, 1)) * np.sin(2 * np.pi * (k + 1) * t + rng.uniform(0, 6.28, (C, 1))) / (k + 1) for k in range(4))
    emask = np.ones(L, dtype=bool)
    emask[::rate] = False
    interp = torch.from_numpy(x.astype(np.float32))
    inp = interp * torch.from_numpy(~emask).view(1, -1)
    return inp, interp, torch.from_numpy(emask), torch.zeros(L, dtype=torch.bool)
