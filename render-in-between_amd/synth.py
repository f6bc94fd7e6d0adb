"""Deterministic synthetic checkpoint and inputs.

The reference ships no checkpoint and a random-init generator with
random spectral-norm vectors is numerically meaningless (sigma is a
tiny random number, activations explode).  ``make_state_dict`` therefore
builds, as a pure function of a seed, a state-dict with the reference's exact
key set and shapes whose ``weight_u / weight_v`` are the converged power
iteration vectors of ``weight_orig`` (what training leaves in a real
checkpoint), so that ``weight_orig / sigma`` has unit spectral norm.

The same function runs on the build box (golden generation, CPU tests) and on
the GPU box (bench, smoke, parity tests): nothing about the weights has to
travel except the seed.
"""
from __future__ import annotations

import hashlib

import numpy as np
import torch

from .config import GenSpec
from .spec import conv_inventory


def _rng(seed: int, idx: int, sub: int):
    return np.random.default_rng([int(seed), int(idx), int(sub)])


def _power_iterate(w_mat: np.ndarray, rng, iters: int = 30):
    """u, v with v = W^T u / |.|, u = W v / |.| (torch.nn.utils.spectral_norm's
    iteration, run to convergence in fp64)."""
    w = w_mat.astype(np.float64)
    u = rng.standard_normal(w.shape[0])
    u /= max(np.linalg.norm(u), 1e-12)
    v = None
    for _ in range(iters):
        v = w.T @ u
        v /= max(np.linalg.norm(v), 1e-12)
        u = w @ v
        u /= max(np.linalg.norm(u), 1e-12)
    return u.astype(np.float32), v.astype(np.float32)


def make_state_dict(spec: GenSpec, seed: int = 0, power_iters: int = 30):
    """Reference-compatible state-dict (torch fp32 CPU tensors)."""
    sd = {}
    for idx, c in enumerate(conv_inventory(spec)):
        k = c.ksize
        fan_in = c.cin * k * k
        if c.spade_cond:
            r = _rng(seed, idx, 0)
            w = r.standard_normal((2 * c.cin, c.spade_cond, 1, 1)) * (0.5 / np.sqrt(c.spade_cond))
            b = r.standard_normal(2 * c.cin) * 0.1
            sd[c.spade_prefix + ".weight"] = w.astype(np.float32)
            sd[c.spade_prefix + ".bias"] = b.astype(np.float32)
        r = _rng(seed, idx, 1)
        w = (r.standard_normal((c.cout, c.cin, k, k)) * np.sqrt(2.0 / fan_in)).astype(np.float32)
        b = (r.standard_normal(c.cout) * 0.05).astype(np.float32)
        if c.spectral:
            u, v = _power_iterate(w.reshape(c.cout, -1), _rng(seed, idx, 2), power_iters)
            sd[c.conv_prefix + ".bias"] = b
            sd[c.conv_prefix + ".weight_orig"] = w
            sd[c.conv_prefix + ".weight_u"] = u
            sd[c.conv_prefix + ".weight_v"] = v
        else:
            sd[c.conv_prefix + ".weight"] = w
            sd[c.conv_prefix + ".bias"] = b
        if c.in_affine:
            r = _rng(seed, idx, 3)
            sd[c.norm_prefix + ".weight"] = (1.0 + 0.1 * r.standard_normal(c.cout)).astype(np.float32)
            sd[c.norm_prefix + ".bias"] = (0.1 * r.standard_normal(c.cout)).astype(np.float32)
    return {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in sd.items()}


def state_dict_digest(sd) -> str:
    """SHA-256 over names, shapes and raw fp32 bytes (fixture identity)."""
    h = hashlib.sha256()
    for k in sorted(sd):
        t = sd[k].detach().cpu().contiguous()
        h.update(k.encode())
        h.update(str(tuple(t.shape)).encode())
        h.update(t.numpy().tobytes())
    return h.hexdigest()


def make_inputs(spec: GenSpec, B: int, H: int, W: int, seed: int = 0, blobs: bool = True):
    """Synthetic (label, img_fake, img_prev) of the shapes and value ranges the
    driver feeds (evaluator.py:228-229,250; HSM_auto_dataset.py:73-75,205-236):
    label channels 0-2 = skeleton image in [-1,1], channels 3.. = joint
    heat-maps in [0,1] (gaussian blobs, sigma 5, peak 1); images in [-1,1]."""
    r = np.random.default_rng([int(seed), 7919])
    label = np.empty((B, spec.label_nc, H, W), np.float32)
    label[:, :3] = r.uniform(-1, 1, (B, 3, H, W))
    nh = spec.label_nc - 3
    if blobs:
        yy, xx = np.mgrid[0:H, 0:W].astype(np.float32)
        for b in range(B):
            for j in range(nh):
                cy, cx = r.uniform(0, H), r.uniform(0, W)
                label[b, 3 + j] = np.exp(-((yy - cy) ** 2 + (xx - cx) ** 2) / (2 * 5.0 ** 2))
    else:
        label[:, 3:] = r.uniform(0, 1, (B, nh, H, W))
    img_fake = r.uniform(-1, 1, (B, spec.image_nc, H, W)).astype(np.float32)
    img_prev = r.uniform(-1, 1, (B, spec.image_nc, H, W)).astype(np.float32)
    return (torch.from_numpy(label), torch.from_numpy(img_fake), torch.from_numpy(img_prev))


def smooth_image(spec: GenSpec, B: int, H: int, W: int, seed: int):
    """Low-frequency image in [-1,1] (a more frame-like input than white noise)."""
    r = np.random.default_rng([int(seed), 104729])
    yy, xx = np.mgrid[0:H, 0:W].astype(np.float32)
    out = np.zeros((B, spec.image_nc, H, W), np.float32)
    for b in range(B):
        for c in range(spec.image_nc):
            acc = np.zeros((H, W), np.float32)
            for _ in range(4):
                fy, fx = r.uniform(0.5, 6, 2)
                ph = r.uniform(0, 2 * np.pi, 2)
                acc += np.sin(2 * np.pi * fy * yy / H + ph[0]) * np.cos(2 * np.pi * fx * xx / W + ph[1])
            out[b, c] = acc / 4.0
    return torch.from_numpy(out)
