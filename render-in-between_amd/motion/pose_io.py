"""Host-side plumbing of stage 1: OpenPose json folders <-> network coordinates.

Mirrors, function for function, what the reference's inference path runs on the host
(HMM/utils/utils.py:82-230: extract_valid_keypoints, select_largest_bb, openpose2motion,
motion2openpose; HMM/datasets/AMASS_dataset.py:221-264,431-465,519-554: get_openpose_data and its
helpers; HMM/models/evaluator.py:175-232: interpolate_openpose, _post_process).  numpy fp64 like the
reference: a few hundred joints per clip, nothing here is worth a kernel.  Checked bit for bit against
the reference's own outputs (tests/golden/motion_pose_*.npz).
"""
from __future__ import annotations

import json
import os
import shutil

import numpy as np
import torch

_DATA = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data")
BODY_JOINTS = list(range(0, 15)) + [19, 22]     # 15 body joints + both big toes (utils.py:141)
ROOT_IDX = 8                                     # mid-hip in the 19-joint layout (AMASS_dataset.py:531)


def extract_valid_keypoints(pts, thres=0.0):
    valid = pts[:, 2] > thres
    if valid.sum() > 5:
        return np.mean(pts[valid, :], axis=0, keepdims=True)
    return np.zeros((1, 3))


def select_largest_bb(people, thres=0.01):
    best_i, best_area = -1, -1
    for i, person in enumerate(people):
        j = np.array(person["pose_keypoints_2d"]).copy().reshape((-1, 3))[:15, :]
        ok = np.where(j[:, 2] > thres)
        if len(ok[0]) < 8:
            continue
        area = (np.amax(j[:, 0][ok]) - np.amin(j[:, 0][ok])) * (np.amax(j[:, 1][ok]) - np.amin(j[:, 1][ok]))
        if area > best_area:
            best_area, best_i = area, i
    return best_i


def openpose2motion(json_dir, scale=None, offset=None, max_frame=None, thres=0.0):
    names = sorted(os.listdir(json_dir))
    names = names[:(max_frame if max_frame is not None else len(names))]
    frames = []
    for path in (os.path.join(json_dir, x) for x in names if x.endswith(".json")):
        with open(path) as f:
            people = json.load(f)["people"]
        idx = select_largest_bb(people) if len(people) > 0 else -1
        if idx != -1:
            person = people[idx]
            body = np.array(person["pose_keypoints_2d"]).reshape(-1, 3)[BODY_JOINTS]
            hands = [extract_valid_keypoints(np.array(person[k]).reshape(-1, 3))
                     for k in ("hand_left_keypoints_2d", "hand_right_keypoints_2d")]
            joints = np.concatenate([body] + hands, axis=0)
            conf = joints[:, 2].copy()
            kept = np.zeros_like(joints)
            kept[conf > thres, :] = joints[conf > thres, :]
            kept[:, 2] = conf
        else:
            # nobody detected: repeat the previous frame (the reference needs two earlier frames for that)
            kept = frames[-1] if len(frames) > 1 else np.zeros((19, 3))
        frames.append(kept)
    motion = np.stack(frames, axis=0)
    conf = motion[:, :, -1]
    valid = conf > thres
    scale = 512 if scale is None else scale
    offset = 256 if offset is None else offset
    motion = (motion[:, :, :2] - offset) / scale
    motion[~valid, :] = 0.0
    return motion.transpose(1, 2, 0), conf[:, :, np.newaxis].transpose(1, 2, 0), (scale, offset)


def motion2openpose(motion, conf, save_json_dir, scale=512.0, offset=256.0, sample_rate=8):
    if not os.path.exists(save_json_dir):
        print("Creating directory: {}".format(save_json_dir))
        os.makedirs(save_json_dir)
    for i in range(motion.shape[-1]):
        joints = motion[:, :, i].copy() * scale + offset
        c = conf[:, :, i].copy()
        body = np.pad(np.concatenate([joints[:15], c[:15]], axis=1), ((0, 10), (0, 0)), "constant", constant_values=0.0)
        body[19, :] = np.concatenate([joints[15], c[15]], axis=None)
        body[22, :] = np.concatenate([joints[16], c[16]], axis=None)

        def hand(j):
            return np.concatenate([joints[j], c[j]], axis=None)[np.newaxis, :].repeat(21, axis=0).reshape(-1).tolist()
        person = {"person_id": [-1], "pose_keypoints_2d": body.reshape(-1).tolist(), "face_keypoints_2d": [],
                  "hand_left_keypoints_2d": hand(17), "hand_right_keypoints_2d": hand(18),
                  "pose_keypoints_3d": [], "face_keypoints_3d": [], "hand_left_keypoints_3d": [], "hand_right_keypoints_3d": []}
        with open(os.path.join(save_json_dir, "{:06d}_keypoints.json".format(i)), "w") as fp:
            json.dump({"version": 1.3, "people": [person]}, fp)


class OpenPoseClips:
    """The slice of AMASSDataset that inference touches: pose statistics + get_openpose_data."""

    def __init__(self, cfg):
        get = cfg.get if isinstance(cfg, dict) else (lambda k, d=None: getattr(cfg, k, d))
        self.openpose_scale = get("openpose_scale", 512)
        self.openpose_offset = get("openpose_offset", 256)
        self.return_type = get("return_type", "network")
        if self.return_type == "3D":
            raise NotImplementedError("return_type '3D' is a training/debug mode; inference uses 2-D joints")
        root = get("data_root", "") or _DATA
        tag = "%s_%s_%.0f_%.0f.npy" % (self.return_type, get("camera_project", "perspective"), get("focal", 4.0), get("depth", 4.0))
        try:
            self.mean_pose = np.load(os.path.join(root, "mean_pose_" + tag)).copy()
            self.std_pose = np.load(os.path.join(root, "std_pose_" + tag)).copy()
        except OSError as e:
            raise ValueError("pose statistics not found under '%s' (%s); the reference would recompute them from the AMASS h5 "
                             "file, which inference does not have" % (root, e))

    @staticmethod
    def _interpolate_frames(data, mask, conf, times):
        for _ in range(times):   # one round: midpoints between neighbours, L -> 2L - 1
            L = data.shape[-1]
            d2 = np.zeros((data.shape[0], data.shape[1], 2 * L - 1)); c2 = np.zeros((conf.shape[0], conf.shape[1], 2 * L - 1))
            m2 = np.zeros(2 * L - 1, dtype=np.int32)
            d2[:, :, ::2] = data; d2[:, :, 1::2] = (data[:, :, 1:] + data[:, :, :-1]) / 2
            c2[:, :, ::2] = conf; c2[:, :, 1::2] = (conf[:, :, 1:] + conf[:, :, :-1]) / 2
            m2[::2] = mask; m2[1::2] = mask[1:]
            data, conf, mask = d2, c2, m2
        return data, mask, conf

    def _localize_motion(self, motion):
        centers = motion[ROOT_IDX, :, :]
        motion = motion - centers
        return np.r_[motion[:ROOT_IDX], motion[ROOT_IDX + 1:], centers[np.newaxis, :, :]]

    def _normalize_motion(self, motion):
        return (motion - self.mean_pose[:, :, np.newaxis]) / self.std_pose[:, :, np.newaxis]

    def get_openpose_data(self, json_dir, sample_rate=8):
        motion, conf, (scale, offset) = openpose2motion(json_dir, scale=self.openpose_scale, offset=self.openpose_offset)
        decoder_mask = np.array([0] * motion.shape[-1])
        run = int(np.log2(sample_rate))
        interp, interp_mask, interp_conf = self._interpolate_frames(motion.copy(), decoder_mask.copy(), conf.copy(), run)
        L = interp_mask.shape[-1]
        assert (L - 1) % sample_rate == 0
        sample_mask = np.ones(L, dtype=np.int32)
        sample_mask[::sample_rate] = 0
        encoder_mask = np.bitwise_or(sample_mask, interp_mask)
        interp = self._normalize_motion(self._localize_motion(interp))
        interp = interp.reshape([-1, interp.shape[-1]])
        inp = interp.copy() * ~encoder_mask.reshape(1, -1).astype(bool)
        return ((scale, offset, interp_conf), torch.from_numpy(inp).float(), torch.from_numpy(interp).float(),
                torch.from_numpy(encoder_mask).bool(), torch.from_numpy(interp_mask).bool())


class Evaluator:
    """Evaluator.interpolate_openpose and its helpers (HMM/models/evaluator.py:175-232)."""

    def __init__(self, cfg):
        self.cfg = cfg
        self.dataset = OpenPoseClips(cfg)
        self.model = None

    def set_model(self, model):
        self.model = model

    def _post_process(self, data, start=0):
        d = data.detach().cpu().numpy()[0].reshape(-1, 2, data.shape[-1])
        d = d * self.dataset.std_pose[:, :, np.newaxis] + self.dataset.mean_pose[:, :, np.newaxis]
        centers = d[-1].copy()
        inv = np.r_[d[:ROOT_IDX], np.zeros((1, 2, d.shape[-1])), d[ROOT_IDX:-1]]
        return inv + centers.reshape((1, 2, -1))

    def interpolate_openpose(self, json_dir, sample_rate, save_dir):
        (scale, offset, conf), input_motion, interp_motion, encoder_mask, decoder_mask = \
            self.dataset.get_openpose_data(json_dir, sample_rate)
        output = self.model.inference(input_motion, interp_motion, encoder_mask, decoder_mask, sample_rate)
        out = self._post_process(output, 0)
        interp = self._post_process(interp_motion.unsqueeze(0), 0)
        for key in ("pred_dir", "linear_dir"):
            if os.path.exists(save_dir[key]):
                shutil.rmtree(save_dir[key])
                print("detete {} ... ".format(save_dir[key]))
        motion2openpose(out, conf, save_dir["pred_dir"], scale=scale, offset=offset, sample_rate=sample_rate)
        motion2openpose(interp, conf, save_dir["linear_dir"], scale=scale, offset=offset, sample_rate=sample_rate)
        print("Inference done!")
        return out, interp
