#!/usr/bin/env python3
"""Pin the one byte-exact op on the path - the output quantisation - to the REFERENCE's own bytes.

Runs the reference's `tensor2images` (PGNR/utils/utils.py:122-147, imported as it is with the easydict /
patoolib stubs of oracle/ref_import.py; build container only) on
  * `fuse_last` of tests/golden/chain3_128.npz (the reference generator's own last fused frame of the 3-step chain),
  * a 1x3x64x64 tensor of knife-edge values: every k/255 grid point mapped back to [-1,1] and its fp32 neighbours
    (where `x*0.5+0.5` lands exactly on, just below and just above a truncation edge), values outside [-1,1],
    +-0, denormals, huge values, +-inf,
asserts that the oracle's restatement (oracle/generator_ref.quantise_uint8) returns the same bytes, and writes the
reference's bytes to tests/golden/quant_ref.npz.  Only numbers are stored.

    python tests/golden/make_golden_quant.py
"""
import importlib.util
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
REF = "/root/reference/Pose_Guided_Neural_Rendering"


def load_reference_utils():
    from oracle import ref_import
    assert ref_import.available(), "reference tree not present"
    if "easydict" not in sys.modules:
        m = types.ModuleType("easydict"); m.EasyDict = ref_import._EasyDict; sys.modules["easydict"] = m
    if "patoolib" not in sys.modules:
        sys.modules["patoolib"] = types.ModuleType("patoolib")
    spec = importlib.util.spec_from_file_location("ref_utils", os.path.join(REF, "utils", "utils.py"))
    ru = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ru)
    return ru


def edge_tensor():
    """[1,3,64,64] fp32: the 256 grid values k/255 as [-1,1] inputs with their fp32 neighbours, and outliers."""
    k = np.arange(256, dtype=np.float64)
    x = ((k / 255.0 - 0.5) / 0.5).astype(np.float32)                 # x*0.5+0.5 ~ k/255
    vals = [x]
    for steps in (1, 2, 3):
        lo, hi = x.copy(), x.copy()
        for _ in range(steps):
            lo = np.nextafter(lo, np.float32(-4)); hi = np.nextafter(hi, np.float32(4))
        vals += [lo, hi]
    mid = (((k + 0.5) / 255.0 - 0.5) / 0.5).astype(np.float32)       # middle of every bucket
    extra = np.array([-1.0, 1.0, -1.0000001, 1.0000001, -2.0, 2.0, 0.0, -0.0, 1e-38, -1e-38, 1e-45, 1e30, -1e30,
                      np.inf, -np.inf, 0.9999999, -0.9999999, 0.99215686, 0.99607843, 0.003921569], np.float32)
    flat = np.concatenate(vals + [mid, extra])
    rng = np.random.default_rng(3)
    n = 3 * 64 * 64
    fill = rng.uniform(-1.2, 1.2, n - flat.size).astype(np.float32)
    return torch.from_numpy(np.concatenate([flat, fill]).reshape(1, 3, 64, 64))


def main():
    from oracle import generator_ref
    ru = load_reference_utils()
    g = np.load(os.path.join(HERE, "chain3_128.npz"))
    fuse_last = torch.from_numpy(g["fuse_last"])
    q_ref = ru.tensor2images(fuse_last)
    assert q_ref.dtype == np.uint8 and q_ref.shape == (128, 128, 3)
    assert np.array_equal(q_ref, generator_ref.quantise_uint8(fuse_last)), "oracle quantiser != reference (chain frame)"
    assert np.array_equal(q_ref, g["quant_last"]), "chain3_128.npz quant_last is not the reference's bytes"
    e = edge_tensor()
    with np.errstate(invalid="ignore"):
        e_ref = ru.tensor2images(e)
        e_orc = generator_ref.quantise_uint8(e)
    assert np.array_equal(e_ref, e_orc), "oracle quantiser != reference (edge values)"
    np.savez_compressed(os.path.join(HERE, "quant_ref.npz"), chain_last_quant=q_ref, edge_in=e.numpy(), edge_quant=e_ref)
    print("quant_ref.npz: chain frame %s, edge tensor %s; oracle == reference bit for bit; %d distinct edge bytes"
          % (q_ref.shape, e_ref.shape, len(np.unique(e_ref))))


if __name__ == "__main__":
    main()
