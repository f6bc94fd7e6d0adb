"""CPU suite: the oracle restatement and the build's spec/synth code against
the fixtures produced from the imported reference (tests/golden/make_golden.py)."""
import json
import os

import numpy as np
import pytest
import torch

import render_in_between_amd as rib
from render_in_between_amd import synth
from oracle import generator_ref

TOL = 2e-5   # oracle(fp32, this torch build) vs stored reference outputs

SMALL_CFG = dict(num_filters=16, max_num_filters=64,
                 mask=dict(num_filters=32, max_num_filters=64),
                 embed=dict(num_filters=32, max_num_filters=64))


def _cfg(name):
    return rib.hsm_gen_config(**SMALL_CFG) if name.startswith("mid") else rib.hsm_gen_config()


def test_state_dict_spec_matches_reference_keys(golden_dir):
    with open(os.path.join(golden_dir, "state_dict_keys.json")) as f:
        ref = {k: tuple(s) for k, s in json.load(f)}
    mine = dict(rib.state_dict_spec(rib.GenSpec.from_cfg(rib.hsm_gen_config())))
    assert len(ref) == 372
    assert set(mine) == set(ref)
    for k in ref:
        assert mine[k] == ref[k], k


def test_conv_flops_match_survey():
    spec = rib.GenSpec.from_cfg(rib.hsm_gen_config())
    assert abs(rib.conv_flops(spec, 512, 512) / 1e9 - 231.60) < 0.01
    assert abs(rib.conv_flops(spec, 256, 256) / 1e9 - 57.90) < 0.01


def test_synth_checkpoint_is_deterministic_and_conditioned(golden_report):
    spec = rib.GenSpec.from_cfg(rib.hsm_gen_config(**SMALL_CFG))
    a = synth.make_state_dict(spec, 7)
    b = synth.make_state_dict(spec, 7)
    assert synth.state_dict_digest(a) == synth.state_dict_digest(b)
    assert synth.state_dict_digest(a) == golden_report["mid_64"]["weights_sha256"]
    # sigma = u.W.v approximates the true spectral norm (power iteration ran)
    p = "down_2.conv_block_0.layers.conv"
    w = a[p + ".weight_orig"].reshape(a[p + ".weight_orig"].shape[0], -1).double()
    sigma = torch.dot(a[p + ".weight_u"].double(), w @ a[p + ".weight_v"].double())
    assert 0.9 < float(sigma) / float(torch.linalg.matrix_norm(w, 2)) <= 1.0 + 1e-6


@pytest.mark.parametrize("name", ["full_64", "full_128", "full_b2_64", "full_noise_128",
                                  "mid_64", "full_256", "full_320x480", "full_b3_96x160"])
def test_oracle_matches_reference_outputs(name, golden_dir, golden_report):
    rep = golden_report[name]
    spec = rib.GenSpec.from_cfg(_cfg(name))
    sd = synth.make_state_dict(spec, rep["seed"])
    assert synth.state_dict_digest(sd) == rep["weights_sha256"]
    label, fake, prev = synth.make_inputs(spec, rep["B"], rep["H"], rep["W"], rep["seed"],
                                          blobs=(name != "full_noise_128"))
    img, mask = generator_ref.RefGenerator(spec, sd)(label, None, fake, prev)
    g = np.load(os.path.join(golden_dir, name + ".npz"))
    s = rep["sub"]
    assert np.abs(img[:, :, ::s, ::s].numpy() - g["img"]).max() <= TOL
    assert np.abs(mask[:, :, ::s, ::s].numpy() - g["mask"]).max() <= TOL
    assert abs(float(img.double().mean()) - rep["img"]["mean"]) < 1e-5
    assert abs(float(mask.double().mean()) - rep["mask"]["mean"]) < 1e-5
    # known-answer invariants
    assert float(img.abs().max()) < 1.0 and 0.0 < float(mask.min()) and float(mask.max()) < 1.0


def test_oracle_taps_match_reference_layers(golden_dir, golden_report):
    rep = golden_report["mid_64"]
    spec = rib.GenSpec.from_cfg(_cfg("mid_64"))
    sd = synth.make_state_dict(spec, rep["seed"])
    label, fake, prev = synth.make_inputs(spec, 1, 64, 64, rep["seed"])
    taps = {}
    generator_ref.RefGenerator(spec, sd)(label, None, fake, prev, taps=taps)
    g = np.load(os.path.join(golden_dir, "mid_64_taps.npz"))
    with open(os.path.join(golden_dir, "mid_64_tap_names.json")) as f:
        pairs = json.load(f)          # reference module name -> oracle tap name
    checked = 0
    for rn, on in pairs.items():
        ref = g[rn.replace(".", "__")]
        mine = taps[on]
        mine = (mine[:, :, ::2, ::2] if mine.shape[-1] >= 32 else mine).numpy()
        assert np.abs(mine - ref).max() <= TOL * max(1.0, np.abs(ref).max()), rn
        checked += 1
    assert checked >= 40


def test_oracle_chain_blend_quantise(golden_dir, golden_report):
    rep = golden_report["chain3_128"]
    spec = rib.GenSpec.from_cfg(rib.hsm_gen_config())
    sd = synth.make_state_dict(spec, rep["seed"])
    H, W = rep["H"], rep["W"]
    key = synth.smooth_image(spec, 1, H, W, 1100)
    labels = [synth.make_inputs(spec, 1, H, W, 1100 + t)[0] for t in range(3)]
    dains = [synth.smooth_image(spec, 1, H, W, 1200 + t) for t in range(3)]
    _, masks, fuses = generator_ref.autoregressive_segment(
        generator_ref.RefGenerator(spec, sd), key, labels, dains)
    g = np.load(os.path.join(golden_dir, "chain3_128.npz"))
    assert np.abs(fuses[0].numpy() - g["fuse0"]).max() <= TOL
    assert np.abs(fuses[-1].numpy() - g["fuse_last"]).max() <= 5 * TOL
    q = generator_ref.quantise_uint8(fuses[-1])
    assert q.dtype == np.uint8 and q.shape == (H, W, 3)
    # truncation can flip a value sitting on an integer boundary; allow 1 LSB on <0.1 %
    diff = np.abs(q.astype(int) - g["quant_last"].astype(int))
    assert diff.max() <= 1 and (diff > 0).mean() < 1e-3


def test_known_answer_properties():
    """label_prev independence (F3), batch independence (F9), determinism."""
    spec = rib.GenSpec.from_cfg(rib.hsm_gen_config(**SMALL_CFG))
    sd = synth.make_state_dict(spec, 3)
    R = generator_ref.RefGenerator(spec, sd)
    label, fake, prev = synth.make_inputs(spec, 2, 32, 48, 3)
    a = R(label, None, fake, prev)
    b = R(label, torch.randn_like(label), fake, prev)
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
    c = R(label[1:], None, fake[1:], prev[1:])
    assert float((a[0][1:] - c[0]).abs().max()) < 1e-5
    assert float((a[1][1:] - c[1]).abs().max()) < 1e-5


def test_unsupported_variants_are_rejected():
    for ov in (dict(kernel_size=5), dict(embed=dict(arch="unet")),
               dict(activation_norm_params=dict(num_filters=64)),
               dict(activation_norm_params=dict(kernel_size=3))):
        with pytest.raises(NotImplementedError):
            rib.GenSpec.from_cfg(rib.hsm_gen_config(**ov))
    assert generator_ref.sample_rate_of(9, 3) == 4 and generator_ref.sample_rate_of(3, 2) == 2


def test_oracle_rasteriser_matches_reference_outputs(golden_dir):
    """oracle/rasterise_ref.py against the outputs of the reference's own functions
    (tests/golden/make_golden_raster.py): json reader, limb drawing, heat-maps - bit for bit."""
    from oracle import rasterise_ref as R
    for n in "abcde":
        g = np.load(os.path.join(golden_dir, "raster_%s.npz" % n))
        H, W = [int(v) for v in g["size"]]
        kp = R.read_json_keypoint(os.path.join(golden_dir, "raster_json", "pose_%s.json" % n))
        assert np.array_equal(kp, g["keypoints"])
        w0, h0 = [int(v) for v in g["orig"]]
        lm = [(kp[i, 0] * (W / w0), kp[i, 1] * (H / h0)) for i in range(19)]
        assert np.array_equal(np.array(lm), g["landmarks"])
        conf = list(kp[:, 2])
        assert np.array_equal(R.skeleton_image(lm, conf, H, W), g["skeleton"]), n
        assert np.array_equal(R.pose_map(lm, conf, H, W), g["pose_map"]), n


def test_quantiser_restatement_equals_the_reference_bytes(golden_dir):
    """The one byte-exact op on the path: oracle quantise_uint8 == the reference's tensor2images output
    (tests/golden/quant_ref.npz, made by make_golden_quant.py from PGNR/utils/utils.py:122-147) bit for bit, on the
    reference's own chain frame and on knife-edge values."""
    import numpy as np
    import torch
    from oracle import generator_ref
    g = np.load(os.path.join(golden_dir, "quant_ref.npz"))
    c = np.load(os.path.join(golden_dir, "chain3_128.npz"))
    assert np.array_equal(generator_ref.quantise_uint8(torch.from_numpy(c["fuse_last"])), g["chain_last_quant"])
    assert np.array_equal(c["quant_last"], g["chain_last_quant"])
    with np.errstate(invalid="ignore"):
        assert np.array_equal(generator_ref.quantise_uint8(torch.from_numpy(g["edge_in"])), g["edge_quant"])
    assert len(np.unique(g["edge_quant"])) == 256
