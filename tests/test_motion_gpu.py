"""GPU parity tests for stage 1, the motion transformer: the HIP path, called through
the C ABI of include/rib_motion.h, against the CPU oracle on the same seeded inputs and against the
committed outputs of the reference itself (tests/golden/motion_*.npz).

Tolerance: fp32 arithmetic on both sides in different summation orders; outputs are O(1..7).  The
oracle agrees with the reference to 4e-6; the bar here is 1e-4 max-abs (measured ~1e-5)."""
import json
import os

import numpy as np
import pytest
import torch

import render_in_between_amd  # noqa: F401
from render_in_between_amd.motion import MotionSpec, synth, pose_io
from oracle import motion_ref

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
TOL = 1e-4


def build(spec, seed):
    from render_in_between_amd.motion import model
    sd = synth.make_state_dict(spec, seed)
    T = model.MotionTransformer(spec, device="cuda:0").eval()
    T.load_state_dict(sd)
    return model, T, sd


@pytest.mark.parametrize("name", ["a", "b", "c", "d"])
def test_hip_transformer_matches_reference_outputs(name):
    z = np.load(os.path.join(GOLDEN, "motion_net_%s.npz" % name))
    cfg = json.loads(str(z["spec"]))
    spec = MotionSpec(**cfg)
    model, T, sd = build(spec, int(z["seed"]))
    P = model.PositionEmbeddingSine1D(spec.pos_hidden_dim // 2)
    src, tgt = torch.from_numpy(z["src"]).cuda(), torch.from_numpy(z["tgt"]).cuda()
    sm, tm = torch.from_numpy(z["src_mask"]).cuda(), torch.from_numpy(z["tgt_mask"]).cuda()
    joints, reco = T(src, sm, P(sm), tgt, tm, P(tm), int(z["rate"]))
    torch.cuda.synchronize()
    # the positional table is the reference's, bit for bit
    assert torch.equal(P(sm).cpu(), motion_ref.position_embedding_sine(sm.cpu(), spec.pos_hidden_dim // 2))
    dj = float((joints.cpu() - torch.from_numpy(z["joints"])).abs().max())
    dr = float((reco.cpu() - torch.from_numpy(z["reco"])).abs().max())
    print("motion net %s: max-abs vs reference joints %.2e reco %.2e" % (name, dj, dr))
    assert dj <= TOL and dr <= TOL
    # and the oracle on the same inputs
    npf = spec.pos_hidden_dim // 2
    oj, orc = motion_ref.transformer_forward(sd, cfg, src.cpu(), sm.cpu(), motion_ref.position_embedding_sine(sm.cpu(), npf), tgt.cpu(),
                                             tm.cpu(), motion_ref.position_embedding_sine(tm.cpu(), npf), int(z["rate"]))
    assert float((joints.cpu() - oj).abs().max()) <= TOL and float((reco.cpu() - orc).abs().max()) <= TOL


@pytest.mark.parametrize("over,N,n_key,rate", [
    (dict(), 2, 3, 16),                                                        # batch of clips, longest key-frame spacing
    (dict(hidden_dim=32, pos_hidden_dim=32, nheads=1, dim_feedforward=40, enc_layers=1, dec_layers=1), 1, 2, 2),   # head_dim 32, L = 3
    (dict(hidden_dim=256, pos_hidden_dim=256, nheads=4, dim_feedforward=1024, enc_layers=1, dec_layers=2, activation="gelu"), 1, 9, 8),  # head_dim 64, widest FFN
    (dict(hidden_dim=64, pos_hidden_dim=64, nheads=8, dim_feedforward=64, enc_layers=2, dec_layers=2, pre_norm=False, activation="relu"), 3, 17, 4),  # head_dim 8, post-norm
    (dict(input_joints=6, two_stage=False), 1, 4, 8),
    (dict(enc_layers=1, dec_layers=1), 1, 176, 8),                             # 1401 frames: score tile exceeds LDS, streamed-key attention kernel
])
def test_hip_transformer_matches_oracle_on_config_variants(over, N, n_key, rate):
    spec = MotionSpec(**over)
    model, T, sd = build(spec, 11)
    P = model.PositionEmbeddingSine1D(spec.pos_hidden_dim // 2)
    clips = [synth.make_clip(spec, n_key, rate, 50 + n) for n in range(N)]
    src = torch.stack([c[0] for c in clips]); tgt = torch.stack([c[1] for c in clips])
    sm = torch.stack([c[2] for c in clips]); tm = torch.stack([c[3] for c in clips])
    if N > 1:
        tm[0, -1] = True
    joints, reco = T(src, sm, P(sm), tgt, tm, P(tm), rate)
    torch.cuda.synchronize()
    npf = spec.pos_hidden_dim // 2
    oj, orc = motion_ref.transformer_forward(sd, spec.as_dict(), src, sm, motion_ref.position_embedding_sine(sm, npf), tgt, tm,
                                             motion_ref.position_embedding_sine(tm, npf), rate)
    dj, dr = float((joints.cpu() - oj).abs().max()), float((reco.cpu() - orc).abs().max())
    print("variant %s N=%d L=%d: joints %.2e reco %.2e" % (over, N, src.shape[-1], dj, dr))
    assert dj <= TOL and dr <= TOL


def test_streamed_key_attention_kernel_agrees_with_the_tiled_one(monkeypatch):
    spec = MotionSpec(enc_layers=2, dec_layers=2)
    model, T, sd = build(spec, 8)
    P = model.PositionEmbeddingSine1D(64)
    src, tgt, sm, tm = [t.unsqueeze(0) for t in synth.make_clip(spec, 12, 8, 3)]
    j1, r1 = T(src, sm, P(sm), tgt, tm, P(tm), 8)
    monkeypatch.setenv("RIBM_NO_TILE_ATTENTION", "1")
    j2, r2 = T(src, sm, P(sm), tgt, tm, P(tm), 8)
    torch.cuda.synchronize()
    assert float((j1 - j2).abs().max()) <= 2e-5 and float((r1 - r2).abs().max()) <= 2e-5


def test_interpolation_between_key_frames_is_bit_exact_and_deterministic():
    spec = MotionSpec()
    model, T, sd = build(spec, 5)
    P = model.PositionEmbeddingSine1D(64)
    src, tgt, sm, tm = [t.unsqueeze(0) for t in synth.make_clip(spec, 6, 8, 1)]
    j1, r1 = T(src, sm, P(sm), tgt, tm, P(tm), 8)
    j2, r2 = T(src, sm, P(sm), tgt, tm, P(tm), 8)
    torch.cuda.synchronize()
    assert torch.equal(j1, j2) and torch.equal(r1, r2)            # no atomics, fixed summation order
    # joints - joints_embed(...) = center = interpolate_embedding(reco): check through a zeroed output head
    sd0 = dict(sd); sd0["joints_embed.weight"] = torch.zeros_like(sd["joints_embed.weight"]); sd0["joints_embed.bias"] = torch.zeros_like(sd["joints_embed.bias"])
    T.load_state_dict(sd0)
    j0, r0 = T(src, sm, P(sm), tgt, tm, P(tm), 8)
    torch.cuda.synchronize()
    assert torch.equal(r0.cpu(), src.permute(2, 0, 1))             # reco = 0 + input
    assert torch.equal(j0.cpu(), motion_ref.interpolate_embedding(r0.cpu(), 8))


def test_fully_masked_keys_give_nan_like_the_reference_and_bad_rate_is_refused():
    spec = MotionSpec(enc_layers=1, dec_layers=1)
    model, T, sd = build(spec, 2)
    P = model.PositionEmbeddingSine1D(64)
    src, tgt, sm, tm = [t.unsqueeze(0) for t in synth.make_clip(spec, 3, 4, 1)]
    all_masked = torch.ones_like(sm)
    j, r = T(src, all_masked, P(sm), tgt, tm, P(tm), 4)
    torch.cuda.synchronize()
    oj, _ = motion_ref.transformer_forward(sd, spec.as_dict(), src, all_masked, motion_ref.position_embedding_sine(sm, 64), tgt, tm,
                                           motion_ref.position_embedding_sine(tm, 64), 4)
    assert torch.isnan(oj).all() and torch.isnan(j).all()
    from render_in_between_amd.motion import _native
    with pytest.raises(_native.RibmError):
        T(src, sm, P(sm), tgt, tm, P(tm), 5)                       # (L - 1) % rate != 0: the reference indexes out of range
    with pytest.raises(RuntimeError):
        T(src[:, :10], sm, P(sm), tgt, tm, P(tm), 4)


@pytest.mark.parametrize("name", ["a", "b"])
def test_stage1_cli_on_openpose_folders_matches_reference(tmp_path, name):
    """The whole stage: json folder -> Predict_motion / Linear_motion json, through the drop-in CLI."""
    import importlib.util
    import yaml
    z = np.load(os.path.join(GOLDEN, "motion_pose_%s.npz" % name))
    spec = MotionSpec()
    ck = tmp_path / "model_epoch399.pth"
    torch.save(synth.make_state_dict(spec, int(z["seed"])), ck)
    cfg = yaml.load(open(os.path.join(ROOT, "render-in-between_amd", "configs", "motion.yaml")), Loader=yaml.FullLoader)
    cfg["model_pretrain"] = str(ck)
    cpath = tmp_path / "motion.yaml"
    yaml.dump(cfg, open(cpath, "w"))
    pose_dir = tmp_path / "poses"
    os.makedirs(pose_dir)
    os.symlink(os.path.join(GOLDEN, "motion_json", name), pose_dir / "clip0")
    s = importlib.util.spec_from_file_location("motion_inference", os.path.join(ROOT, "render-in-between_amd", "motion", "inference.py"))
    cli = importlib.util.module_from_spec(s); s.loader.exec_module(cli)
    import argparse
    cli.main(argparse.Namespace(config=str(cpath), save_dir=str(tmp_path / "out"), pose_dir=str(pose_dir), upsample_rate=int(z["rate"]), seed=123))
    files = [str(f) for f in z["files"]]
    pred_dir = tmp_path / "out" / "Predict_motion" / "clip0"
    assert sorted(os.listdir(pred_dir)) == files and sorted(os.listdir(tmp_path / "out" / "Linear_motion" / "clip0")) == files
    motion, conf, _ = pose_io.openpose2motion(str(pred_dir), scale=512, offset=256)
    # written joints (image pixels / 512): the reference's own post-processed prediction
    ref = z["out"]
    got = motion * 1.0
    keep = [i for i in range(19)]
    d = np.abs(got[keep] - ref[keep]) * (z["conf"][keep] > 0)
    print("stage 1 folder %s: max-abs joint error %.2e (network units)" % (name, d.max()))
    assert d.max() <= 1e-4
    lin = pose_io.openpose2motion(str(tmp_path / "out" / "Linear_motion" / "clip0"), scale=512, offset=256)[0]
    assert (np.abs(lin - z["linear"]) * (z["conf"] > 0)).max() <= 1e-12


def test_full_length_clip_ignores_what_is_stored_at_non_key_frames():
    """Size-independent property at the reference's full clip length (321 frames): with two_stage the prediction
    depends on the key frames only - non-key frames are masked as attention keys and interpolate_embedding reads
    multiples of `rate` - so garbage at the masked positions of the encoder input must not change a bit of it."""
    spec = MotionSpec()
    model, T, sd = build(spec, 4)
    P = model.PositionEmbeddingSine1D(64)
    src, tgt, sm, tm = [t.unsqueeze(0) for t in synth.make_clip(spec, 41, 8, 9)]
    j1, r1 = T(src, sm, P(sm), tgt, tm, P(tm), 8)
    noisy = src.clone()
    noisy[:, :, sm[0]] = torch.randn(1, spec.input_joints, int(sm.sum())) * 10.0
    j2, r2 = T(noisy, sm, P(sm), None, tm, P(tm), 8)
    torch.cuda.synchronize()
    assert torch.equal(j1, j2)
    assert torch.equal(r1[::8], r2[::8])                   # the reconstruction at the key frames too
    assert not torch.equal(r1, r2)                         # (elsewhere it is input + correction, so it differs)
