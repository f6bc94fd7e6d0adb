"""Generate the golden fixtures in this directory.

Run ONLY in the build container (needs /root/reference):

    python tests/golden/make_golden.py

For every case it (1) builds the seed-defined synthetic checkpoint and inputs
with the build's own code (render_in_between_amd.synth), (2) runs the
*imported reference generator* (oracle/ref_import.py) on them, (3) asserts the
oracle restatement (oracle/generator_ref.py) agrees with the reference to
<= 1e-5 max-abs, and (4) stores the REFERENCE's outputs.  Only numbers are
stored; no reference source text.
"""
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

import render_in_between_amd as rib                      # noqa: E402
from render_in_between_amd import synth                  # noqa: E402
from oracle import generator_ref, ref_import             # noqa: E402

ORACLE_TOL = 1e-5

# a narrow variant of HSM.yaml's gen block that keeps every structural feature (learned and
# identity shortcuts, 5 cond levels, stride-2 encoders) with at most 64 channels
SMALL_CFG = dict(num_filters=16, max_num_filters=64,
                 mask=dict(num_filters=32, max_num_filters=64),
                 embed=dict(num_filters=32, max_num_filters=64))


def summary(t):
    t = t.double()
    return {"mean": float(t.mean()), "abs_mean": float(t.abs().mean()),
            "min": float(t.min()), "max": float(t.max()),
            "sumsq": float((t * t).sum())}


def run_case(name, cfg, seed, B, H, W, sub, report, blobs=True, tol=ORACLE_TOL):
    spec = rib.GenSpec.from_cfg(cfg)
    sd = synth.make_state_dict(spec, seed)
    G = ref_import.load_reference_generator(cfg)
    G.load_state_dict(sd, strict=True)
    label, fake, prev = synth.make_inputs(spec, B, H, W, seed, blobs=blobs)
    with torch.no_grad():
        rimg, rmask = G(label, torch.zeros_like(label), fake, prev)
        # label_prev is dead
        rimg2, _ = G(label, None, fake, prev)
    assert torch.equal(rimg, rimg2)
    oimg, omask = generator_ref.RefGenerator(spec, sd)(label, None, fake, prev)
    d_img = float((rimg - oimg).abs().max())
    d_mask = float((rmask - omask).abs().max())
    assert d_img <= tol and d_mask <= tol, (name, d_img, d_mask)
    report[name] = {"B": B, "H": H, "W": W, "seed": seed, "sub": sub,
                    "oracle_vs_reference": {"img": d_img, "mask": d_mask},
                    "weights_sha256": synth.state_dict_digest(sd),
                    "img": summary(rimg), "mask": summary(rmask)}
    np.savez_compressed(os.path.join(HERE, name + ".npz"),
                        img=rimg[:, :, ::sub, ::sub].numpy(),
                        mask=rmask[:, :, ::sub, ::sub].numpy())
    print("%-16s oracle-vs-ref img %.2e mask %.2e" % (name, d_img, d_mask))
    return spec, sd, G


def main():
    torch.manual_seed(0)
    torch.set_num_threads(8)
    report = {}
    full = rib.hsm_gen_config()

    # state-dict key/shape pin
    G = ref_import.load_reference_generator(full)
    keys = [[k, list(v.shape)] for k, v in G.state_dict().items()]
    with open(os.path.join(HERE, "state_dict_keys.json"), "w") as f:
        json.dump(keys, f, indent=0)

    run_case("full_64", full, 0, 1, 64, 64, 1, report)
    run_case("full_128", full, 1, 1, 128, 128, 1, report)
    run_case("full_256", full, 2, 1, 256, 256, 4, report)
    run_case("full_320x480", full, 3, 1, 320, 480, 8, report)
    run_case("full_512", full, 0, 1, 512, 512, 8, report)
    run_case("full_b2_64", full, 4, 2, 64, 64, 1, report)
    run_case("full_noise_128", full, 5, 1, 128, 128, 2, report, blobs=False)

    # shrunken config with per-layer taps from forward hooks on the reference
    small = rib.hsm_gen_config(**SMALL_CFG)
    spec, sd, G = run_case("mid_64", small, 7, 1, 64, 64, 1, report)
    label, fake, prev = synth.make_inputs(spec, 1, 64, 64, 7)
    taps = {}
    hooks = []
    want = (["down_first", "conv_img"] + ["ref_embedding.conv_first"]
            + ["ref_embedding.down_%d" % i for i in range(spec.emb_down)]
            + ["down_%d" % i for i in range(spec.num_down_img + 1)]
            + ["res_%d" % i for i in range(spec.num_res_blocks)]
            + ["up_%d" % i for i in range(spec.num_down_img + 1)]
            + ["flow_network_temp.res_flow.%d" % i for i in range(spec.mask_res_blocks)]
            + ["flow_network_temp.down_lbl", "flow_network_temp.down_img",
               "flow_network_temp.up_flow"]
            + ["%s_%d.conv_block_0" % (k, i) for k in ("down", "up") for i in range(spec.num_down_img + 1)]
            + ["flow_network_temp.down_lbl.%d.layers.conv" % i for i in range(spec.mask_down)]
            + ["flow_network_temp.down_img.%d.layers.conv" % i for i in range(spec.mask_down)]
            + ["flow_network_temp.up_flow.%d.layers.conv" % (2 * j + 1) for j in range(spec.mask_down)])
    mods = dict(G.named_modules())
    for n in want:
        hooks.append(mods[n].register_forward_hook(
            lambda m, i, o, n=n: taps.__setitem__(n, o.detach().clone())))
    with torch.no_grad():
        G(label, None, fake, prev)
    for h in hooks:
        h.remove()
    otaps = {}
    generator_ref.RefGenerator(spec, sd)(label, None, fake, prev, taps=otaps)
    pairs = {"down_first": "down_first", "ref_embedding.conv_first": "cond_0"}
    for i in range(spec.emb_down):
        pairs["ref_embedding.down_%d" % i] = "cond_%d" % (i + 1)
    for n in want:
        if n.startswith(("down_", "res_", "up_")) and n != "down_first":
            pairs[n] = n
    for i in range(spec.mask_res_blocks):
        pairs["flow_network_temp.res_flow.%d" % i] = "mask.res_%d" % i
    pairs["flow_network_temp.up_flow"] = "mask.up_%d" % (spec.mask_down - 1)
    for k in ("down", "up"):
        for i in range(spec.num_down_img + 1):
            pairs["%s_%d.conv_block_0" % (k, i)] = "%s_%d.h" % (k, i)
    for i in range(spec.mask_down):
        pairs["flow_network_temp.down_lbl.%d.layers.conv" % i] = "mask.lbl_%d.raw" % i
        pairs["flow_network_temp.down_img.%d.layers.conv" % i] = "mask.img_%d.raw" % i
    for j in range(spec.mask_down):
        pairs["flow_network_temp.up_flow.%d.layers.conv" % (2 * j + 1)] = "mask.up_%d.raw" % j
    with open(os.path.join(HERE, "mid_64_tap_names.json"), "w") as f:
        json.dump(pairs, f, indent=0, sort_keys=True)
    worst = 0.0
    for rn, on in pairs.items():
        d = float((taps[rn] - otaps[on]).abs().max())
        worst = max(worst, d)
        assert d <= ORACLE_TOL * max(1.0, float(taps[rn].abs().max())), (rn, d)
    report["mid_64"]["taps_oracle_vs_reference_max"] = worst
    # maps of 32x32 and larger are stored at every 2nd pixel to keep the fixture small
    np.savez_compressed(os.path.join(HERE, "mid_64_taps.npz"),
                        **{k.replace(".", "__"): (v[:, :, ::2, ::2] if v.shape[-1] >= 32 else v).numpy()
                           for k, v in taps.items()})

    # 3-step autoregressive chain @128 (driver semantics, evaluator.py:238-266),
    # driven through the REFERENCE generator
    spec = rib.GenSpec.from_cfg(full)
    sd = synth.make_state_dict(spec, 11)
    G = ref_import.load_reference_generator(full)
    G.load_state_dict(sd, strict=True)
    H = W = 128
    key = synth.smooth_image(spec, 1, H, W, 1100)
    labels, dains = [], []
    for t in range(3):
        lab, _, _ = synth.make_inputs(spec, 1, H, W, 1100 + t)
        labels.append(lab)
        dains.append(synth.smooth_image(spec, 1, H, W, 1200 + t))
    with torch.no_grad():
        _, _, rfuse = generator_ref.autoregressive_segment(
            lambda a, b, c, d: G(a, b, c, d), key, labels, dains)
    _, _, ofuse = generator_ref.autoregressive_segment(
        generator_ref.RefGenerator(spec, sd), key, labels, dains)
    d = max(float((a - b).abs().max()) for a, b in zip(rfuse, ofuse))
    assert d <= 5 * ORACLE_TOL, d
    q = generator_ref.quantise_uint8(rfuse[-1])
    report["chain3_128"] = {"H": H, "W": W, "seed": 11, "steps": 3,
                            "oracle_vs_reference": d,
                            "weights_sha256": synth.state_dict_digest(sd),
                            "fuse_last": summary(rfuse[-1])}
    np.savez_compressed(os.path.join(HERE, "chain3_128.npz"),
                        fuse0=rfuse[0].numpy(), fuse_last=rfuse[-1].numpy(), quant_last=q)
    print("chain3_128       oracle-vs-ref %.2e" % d)

    with open(os.path.join(HERE, "golden_report.json"), "w") as f:
        json.dump(report, f, indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
