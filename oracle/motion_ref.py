"""TEST INFRASTRUCTURE (checker only; see oracle/__init__.py).

CPU restatement of stage 1 of the pipeline: the motion transformer that turns
low-frame-rate OpenPose key frames into the interpolated pose sequence the generator is conditioned
on.  ``HMM`` = /root/reference/Human_Motion_Modelling.  Plain PyTorch fp32 functional ops for the
network, numpy (fp64, as the reference) for the json <-> network-coordinate plumbing.

Pinned against the imported reference (tests/golden/make_golden_motion.py, run in the build
container): HMM/models/transformer.py and position_encoding.py import as they are (torch only);
utils/utils.py imports with the easydict / patoolib stubs of oracle/ref_import.py; the dataset
module needs h5py, which this image lacks, so the METHODS used by inference are compiled out of
datasets/AMASS_dataset.py's syntax tree and run unchanged.  Fixtures: tests/golden/motion_*.npz.

Third-party arithmetic: torch.nn.MultiheadAttention / LayerNorm / Linear live in PyTorch (torch
2.10 here; the reference's README pins torch 1.4 for this stage).  ``multihead_attention`` below
restates ``torch.nn.functional.multi_head_attention_forward`` for the case the reference uses
(separate q/k/v inputs, need_weights=True, eval mode) and is pinned against the module itself.
"""
from __future__ import annotations

import json
import math
import os
from typing import Dict, Optional

import numpy as np
import torch
import torch.nn.functional as F

LN_EPS = 1e-5          # nn.LayerNorm default (HMM/models/transformer.py:214-215)
LEAKY_SLOPE = 0.01     # F.leaky_relu default (HMM/models/transformer.py:373-374)


# --------------------------------------------------------------------------
# network
# --------------------------------------------------------------------------
def position_embedding_sine(mask: torch.Tensor, num_pos_feats: int, temperature: float = 10000.0) -> torch.Tensor:
    """PositionEmbeddingSine_1D.forward (HMM/models/position_encoding.py:25-50), normalize=True,
    scale=2*pi.  mask [N][L] (only its shape matters) -> [L][N][2*num_pos_feats]."""
    N, L = mask.shape
    position = torch.arange(0, L, dtype=torch.float32).unsqueeze(0).repeat(N, 1)
    position = position / (position[:, -1:] + 1e-6) * (2 * math.pi)
    dim_t = torch.arange(num_pos_feats, dtype=torch.float32)
    dim_t = temperature ** (2 * (dim_t // 2) / num_pos_feats)
    pe = torch.zeros(N, L, num_pos_feats * 2)
    pe[:, :, 0::2] = torch.sin(position[:, :, None] / dim_t)
    pe[:, :, 1::2] = torch.cos(position[:, :, None] / dim_t)
    return pe.permute(1, 0, 2)


def activation(x: torch.Tensor, kind: str) -> torch.Tensor:
    """_get_activation_fn (HMM/models/transformer.py:365-375)."""
    if kind == "relu":
        return F.relu(x)
    if kind == "gelu":
        return F.gelu(x)
    if kind == "leaky_relu":
        return F.leaky_relu(x, LEAKY_SLOPE)
    raise RuntimeError("activation should be relu/gelu/leaky_relu, not %s" % kind)


def multihead_attention(q_in, k_in, v_in, sd, prefix, nhead, attn_mask=None, key_padding_mask=None):
    """nn.MultiheadAttention.forward in eval mode for [L][N][E] inputs with distinct q/k/v sources
    (HMM/models/transformer.py:227-228,305-311): per-source input projections with the three slices
    of in_proj_weight, q scaled by sqrt(1/head_dim), boolean masks (True = may not attend) turned
    into -inf, softmax over keys, out_proj."""
    Lq, N, E = q_in.shape
    Lk = k_in.shape[0]
    hd = E // nhead
    w, b = sd[prefix + ".in_proj_weight"], sd[prefix + ".in_proj_bias"]
    q = F.linear(q_in, w[:E], b[:E])
    k = F.linear(k_in, w[E:2 * E], b[E:2 * E])
    v = F.linear(v_in, w[2 * E:], b[2 * E:])
    q = q.reshape(Lq, N * nhead, hd).transpose(0, 1)
    k = k.reshape(Lk, N * nhead, hd).transpose(0, 1)
    v = v.reshape(Lk, N * nhead, hd).transpose(0, 1)
    bias = torch.zeros(N, 1, Lq, Lk)
    if attn_mask is not None:
        bias = bias.masked_fill(attn_mask.view(1, 1, Lq, Lk), float("-inf"))
    if key_padding_mask is not None:
        bias = bias.masked_fill(key_padding_mask.view(N, 1, 1, Lk), float("-inf"))
    bias = bias.expand(N, nhead, Lq, Lk).reshape(N * nhead, Lq, Lk)
    scores = torch.baddbmm(bias, q * math.sqrt(1.0 / hd), k.transpose(1, 2))
    attn = torch.softmax(scores, dim=-1)
    out = torch.bmm(attn, v).transpose(0, 1).reshape(Lq, N, E)
    return F.linear(out, sd[prefix + ".out_proj.weight"], sd[prefix + ".out_proj.bias"])


def layer_norm(x, sd, prefix):
    return F.layer_norm(x, (x.shape[-1],), sd[prefix + ".weight"], sd[prefix + ".bias"], LN_EPS)


def encoder_layer(x, sd, p, cfg, attn_mask, key_padding_mask, pos):
    """TransformerEncoderLayer.forward_pre / forward_post (HMM/models/transformer.py:222-254)."""
    if cfg["pre_norm"]:
        x2 = layer_norm(x, sd, p + ".norm1")
        qk = x2 + pos
        x = x + multihead_attention(qk, qk, x2, sd, p + ".self_attn", cfg["nheads"], attn_mask, key_padding_mask)
        x2 = layer_norm(x, sd, p + ".norm2")
        x2 = F.linear(activation(F.linear(x2, sd[p + ".linear1.weight"], sd[p + ".linear1.bias"]), cfg["activation"]),
                      sd[p + ".linear2.weight"], sd[p + ".linear2.bias"])
        return x + x2
    qk = x + pos
    x = x + multihead_attention(qk, qk, x, sd, p + ".self_attn", cfg["nheads"], attn_mask, key_padding_mask)
    x = layer_norm(x, sd, p + ".norm1")
    x2 = F.linear(activation(F.linear(x, sd[p + ".linear1.weight"], sd[p + ".linear1.bias"]), cfg["activation"]),
                  sd[p + ".linear2.weight"], sd[p + ".linear2.bias"])
    return layer_norm(x + x2, sd, p + ".norm2")


def decoder_layer(t, memory, sd, p, cfg, tgt_kpm, mem_kpm, pos, query_pos):
    """TransformerDecoderLayer.forward_pre / forward_post (HMM/models/transformer.py:295-346); the
    reference passes no causal mask (transformer.py:124-132)."""
    H = cfg["nheads"]

    def ffn(x):
        return F.linear(activation(F.linear(x, sd[p + ".linear1.weight"], sd[p + ".linear1.bias"]), cfg["activation"]),
                        sd[p + ".linear2.weight"], sd[p + ".linear2.bias"])
    if cfg["pre_norm"]:
        t2 = layer_norm(t, sd, p + ".norm1")
        qk = t2 + query_pos
        t = t + multihead_attention(qk, qk, t2, sd, p + ".self_attn", H, None, tgt_kpm)
        t2 = layer_norm(t, sd, p + ".norm2")
        t = t + multihead_attention(t2 + query_pos, memory + pos, memory, sd, p + ".multihead_attn", H, None, mem_kpm)
        t2 = layer_norm(t, sd, p + ".norm3")
        return t + ffn(t2)
    qk = t + query_pos
    t = layer_norm(t + multihead_attention(qk, qk, t, sd, p + ".self_attn", H, None, tgt_kpm), sd, p + ".norm1")
    t = layer_norm(t + multihead_attention(t + query_pos, memory + pos, memory, sd, p + ".multihead_attn", H, None, mem_kpm),
                   sd, p + ".norm2")
    return layer_norm(t + ffn(t), sd, p + ".norm3")


def interpolate_embedding(x: torch.Tensor, rate: int) -> torch.Tensor:
    """Transformer.interpolate_embedding (HMM/models/transformer.py:59-75): every frame becomes the
    linear blend of the two key frames (multiples of ``rate``) around it."""
    L = x.shape[0]
    idx = torch.arange(L)
    chunk, remain = idx // rate, idx % rate
    prev = x[chunk * rate]
    nxt = torch.cat([x[(chunk[:-1] + 1) * rate], x[-1].unsqueeze(0)], dim=0)
    r = remain.view(-1, 1, 1)
    return (prev / rate * (rate - r)) + (nxt / rate * r)


def transformer_forward(sd: Dict[str, torch.Tensor], cfg: dict, src, src_mask, src_pos, tgt, tgt_mask, tgt_pos, rate: int):
    """Transformer.forward (HMM/models/transformer.py:78-111).  src/tgt [N][C][L], masks bool [N][L]
    (True = padded), pos [L][N][D] -> (joints [L][N][C], reco [L][N][C])."""
    src = src.permute(2, 0, 1)
    tgt = tgt.permute(2, 0, 1)
    L = src.shape[0]
    x = F.linear(src, sd["input_embed.weight"], sd["input_embed.bias"])
    eye = torch.eye(L).bool()              # encode(): a frame may not attend to itself (transformer.py:113-119)
    for i in range(cfg["enc_layers"]):
        x = encoder_layer(x, sd, "encoder.layers.%d" % i, cfg, eye, src_mask, src_pos)
    if cfg["pre_norm"]:
        x = layer_norm(x, sd, "encoder.norm")
    mem = x
    reco = F.linear(mem, sd["joints_embed.weight"], sd["joints_embed.bias"]) + src
    center = interpolate_embedding(reco, rate) if cfg["two_stage"] else tgt
    t = F.linear(center, sd["input_embed.weight"], sd["input_embed.bias"])
    for i in range(cfg["dec_layers"]):
        t = decoder_layer(t, mem, sd, "decoder.layers.%d" % i, cfg, tgt_mask, src_mask, src_pos, tgt_pos)
    t = layer_norm(t, sd, "decoder.norm")
    joints = F.linear(t, sd["joints_embed.weight"], sd["joints_embed.bias"]) + center
    return joints, reco


def model_inference(sd, cfg, data, interp, encoder_mask, decoder_mask, rate):
    """Model_inference.inference (HMM/inference.py:20-41): one clip, [C][L] inputs -> pred [1][C][L]."""
    src = data.unsqueeze(0)
    tgt = interp.unsqueeze(0)
    sm = encoder_mask.unsqueeze(0)
    tm = decoder_mask.unsqueeze(0)
    npf = cfg["pos_hidden_dim"] // 2
    pred, _ = transformer_forward(sd, cfg, src, sm, position_embedding_sine(sm, npf), tgt, tm,
                                  position_embedding_sine(tm, npf), rate)
    return pred.permute(1, 2, 0)


# --------------------------------------------------------------------------
# OpenPose json <-> network coordinates (numpy, fp64 like the reference)
# --------------------------------------------------------------------------
def extract_valid_keypoints(pts, thres=0.0):
    """HMM/utils/utils.py:82-89: mean of the confident hand points (needs more than 5), else zeros."""
    out = np.zeros((1, 3))
    valid = pts[:, 2] > thres
    if valid.sum() > 5:
        out = np.mean(pts[valid, :], axis=0, keepdims=True)
    return out


def select_largest_bb(people, thres=0.01):
    """HMM/utils/utils.py:91-115: the person whose confident first-15 joints span the largest box."""
    target, best = -1, -1
    for i, person in enumerate(people):
        j = np.array(person["pose_keypoints_2d"]).reshape((-1, 3))[:15, :]
        valid = j[:, 2] > thres
        if valid.sum() < 8:
            continue
        area = (np.amax(j[valid, 0]) - np.amin(j[valid, 0])) * (np.amax(j[valid, 1]) - np.amin(j[valid, 1]))
        if area > best:
            best, target = area, i
    return target


def openpose2motion(json_dir, scale=512, offset=256, thres=0.0):
    """HMM/utils/utils.py:117-177: 19 joints (15 body + both feet + mean of each hand) x (x, y) per
    json file -> motion [19][2][L] in ((p - offset) / scale) coordinates, conf [19][1][L]."""
    files = sorted(os.listdir(json_dir))
    files = [os.path.join(json_dir, x) for x in files if x.endswith(".json")]
    motion = []
    for path in files:
        with open(path) as f:
            jd = json.load(f)
        idx = select_largest_bb(jd["people"]) if len(jd["people"]) > 0 else -1
        if idx != -1:
            sel = list(range(0, 15)) + [19, 22]
            pts = np.array(jd["people"][idx]["pose_keypoints_2d"]).reshape(-1, 3)[sel]
            lp = extract_valid_keypoints(np.array(jd["people"][idx]["hand_left_keypoints_2d"]).reshape(-1, 3))
            rp = extract_valid_keypoints(np.array(jd["people"][idx]["hand_right_keypoints_2d"]).reshape(-1, 3))
            joints = np.concatenate((pts, lp, rp), axis=0)
            conf = joints[:, 2].copy()
            valid = conf > thres
            nj = np.zeros_like(joints)
            nj[valid, :] = joints[valid, :]
            nj[:, 2] = conf
        else:
            nj = motion[-1] if len(motion) > 1 else np.zeros((19, 3))
        motion.append(nj)
    motion = np.stack(motion, axis=0)
    conf = motion[:, :, -1]
    valid = conf > thres
    motion = (motion[:, :, :2] - offset) / scale
    motion[~valid, :] = 0.0
    return motion.transpose(1, 2, 0), conf[:, :, np.newaxis].transpose(1, 2, 0), (scale, offset)


def interpolate_frames(data, mask, conf, times):
    """AMASSDataset._interpolate_frames (HMM/datasets/AMASS_dataset.py:431-465): ``times`` rounds of
    midpoint insertion, L -> 2L-1 each."""
    for _ in range(times):
        L = data.shape[-1]
        nd = np.zeros((data.shape[0], data.shape[1], L * 2 - 1))
        nd[:, :, ::2] = data
        nd[:, :, 1::2] = (data[:, :, 1:] + data[:, :, :-1]) / 2
        nc = np.zeros((conf.shape[0], conf.shape[1], L * 2 - 1))
        nc[:, :, ::2] = conf
        nc[:, :, 1::2] = (conf[:, :, 1:] + conf[:, :, :-1]) / 2
        nm = np.zeros(L * 2 - 1, dtype=np.int32)
        nm[::2] = mask
        nm[1::2] = mask[1:]
        data, conf, mask = nd, nc, nm
    return data, mask, conf


def localize_motion(motion, root_idx=8):
    """AMASSDataset._localize_motion, 2-D branch (AMASS_dataset.py:519-550): joints relative to the
    hip, the hip row dropped, the hip trajectory appended as the last row."""
    centers = motion[root_idx, :, :]
    motion = motion - centers
    return np.r_[motion[:root_idx], motion[root_idx + 1:], centers[np.newaxis, :, :]]


def get_openpose_data(json_dir, sample_rate, mean_pose, std_pose, scale=512, offset=256):
    """AMASSDataset.get_openpose_data (AMASS_dataset.py:240-264) -> ((scale, offset, conf), input
    [38][L] f32, interp [38][L] f32, encoder_mask bool [L], decoder_mask bool [L])."""
    motion, conf, (scale, offset) = openpose2motion(json_dir, scale=scale, offset=offset)
    dmask = np.array([0] * motion.shape[-1])
    run = int(np.log2(sample_rate))
    im, imask, iconf = interpolate_frames(motion, dmask, conf, run)
    L = imask.shape[-1]
    assert (L - 1) % sample_rate == 0
    smask = np.ones(L, dtype=np.int32)
    smask[::sample_rate] = 0
    emask = np.bitwise_or(smask, imask)                       # generate_training_mask (AMASS_dataset.py:221-238)
    im = localize_motion(im)
    im = (im - mean_pose[:, :, np.newaxis]) / std_pose[:, :, np.newaxis]
    im = im.reshape([-1, im.shape[-1]])
    inp = im.copy() * ~emask.reshape(1, -1).astype(bool)
    return ((scale, offset, iconf), torch.from_numpy(inp).float(), torch.from_numpy(im).float(),
            torch.from_numpy(emask).bool(), torch.from_numpy(imask).bool())


def post_process(data, mean_pose, std_pose):
    """Evaluator._post_process + _denormalize + _globalize, 2-D branch (HMM/models/evaluator.py:203-232):
    [1][38][L] network output -> [19][2][L] image-plane joints (hip re-inserted at row 8)."""
    d = data.detach().cpu().numpy()[0].reshape(-1, 2, data.shape[-1])
    d = d * std_pose[:, :, np.newaxis] + mean_pose[:, :, np.newaxis]
    centers = d[-1].copy()
    inv = np.r_[d[:8], np.zeros((1, 2, d.shape[-1])), d[8:-1]]
    return inv + centers.reshape((1, 2, -1))


def motion_to_openpose_dicts(motion, conf, scale=512.0, offset=256.0):
    """motion2openpose (HMM/utils/utils.py:180-230) without the file writes: one json-able dict per frame."""
    out = []
    for i in range(motion.shape[-1]):
        joints = motion[:, :, i].copy() * scale + offset
        c = conf[:, :, i].copy()
        body = np.concatenate([joints[:15], c[:15]], axis=1)
        body = np.pad(body, ((0, 10), (0, 0)), "constant", constant_values=0.0)
        body[19, :] = np.concatenate([joints[15], c[15]], axis=None)
        body[22, :] = np.concatenate([joints[16], c[16]], axis=None)
        person = {"person_id": [-1], "pose_keypoints_2d": body.reshape(-1).tolist(), "face_keypoints_2d": [],
                  "hand_left_keypoints_2d": np.concatenate([joints[17], c[17]], axis=None)[np.newaxis, :].repeat(21, axis=0).reshape(-1).tolist(),
                  "hand_right_keypoints_2d": np.concatenate([joints[18], c[18]], axis=None)[np.newaxis, :].repeat(21, axis=0).reshape(-1).tolist(),
                  "pose_keypoints_3d": [], "face_keypoints_3d": [], "hand_left_keypoints_3d": [], "hand_right_keypoints_3d": []}
        out.append({"version": 1.3, "people": [person]})
    return out
