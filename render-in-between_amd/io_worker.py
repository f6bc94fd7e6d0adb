"""File-side work of the folder driver as plain functions of plain arguments (numpy / PIL only, no torch, no GPU): PNG /
JPEG decode + resize, OpenPose json -> rasteriser tables, PNG encode.  The Evaluator runs them either on threads of its
own process or - the default for the native path - in a pool of worker PROCESSES (forkserver): at several hundred
frames/s the PIL / scipy / json work of 32+ threads keeps the interpreter lock busy enough to starve the thread that
enqueues the GPU work (profiles/r04_driver.jsonl: the launch thread spent 0.1-0.3 s per clip waiting for the lock).

Reference semantics: PGNR/models/evaluator.py:205-235 (per-frame pre-load), PGNR/utils/utils.py:129-142 (save)."""
from __future__ import annotations

import numpy as np

from . import rasterise
from .resize import resize_cubic_u8


def decode_resized_u8(path, width, height, resize="cv2"):
    """PIL decode -> RGB uint8 HWC at the model size, and the file's own (width, height).  `resize="cv2"`: OpenCV's 8-bit
    INTER_CUBIC restated in resize.py (what the reference's albumentations A.Resize(interpolation=cv2.INTER_CUBIC)
    computes: A = -0.75, no low-pass on reduction; unpinned, cv2 is not in this image); "pil": PIL's BICUBIC."""
    from PIL import Image
    img = Image.open(path).convert("RGB")
    w0, h0 = img.size
    if (w0, h0) == (width, height):
        return np.asarray(img, dtype=np.uint8), (w0, h0)
    if resize == "pil":
        return np.asarray(img.resize((width, height), Image.BICUBIC), dtype=np.uint8), (w0, h0)
    return resize_cubic_u8(np.asarray(img, dtype=np.uint8), width, height), (w0, h0)


def normalised_chw(u8):
    """ToTensor + Normalize(0.5, 0.5) (HSM_auto_dataset.py:73-75) of a uint8 HWC frame: float32 CHW in [-1, 1]."""
    a = u8.astype(np.float32) / 255.0
    return np.ascontiguousarray(((a - 0.5) / 0.5).transpose(2, 0, 1))


def image_size(path):
    from PIL import Image
    with Image.open(path) as im:             # header only
        return im.size


def scaled_pose(json_path, orig_size, width, height):
    """json -> (landmarks, conf) in model-size pixels: the keypoints follow the image resize (A.Resize keypoint rule,
    evaluator.py:24-26,219)."""
    pose = rasterise.read_json_keypoint(json_path)
    sx, sy = width / orig_size[0], height / orig_size[1]
    lm = [(pose[i, 0] * sx, pose[i, 1] * sy) for i in range(pose.shape[0])]
    return lm, [pose[i, 2] for i in range(pose.shape[0])]


def load_frame(dain_path, ref_img_path, pose_path, is_key, want_tables, width, height, resize, thres1, thres2):
    """Everything the native pipeline needs of frame i from disk (evaluator.py:205-235):
    (DAIN frame uint8 HWC, key frame float32 CHW in [-1,1] or None, rasteriser tables or (landmarks, conf)).
    The "gt" image of a frame is its segment's key frame (or gt_dir's frame i); the keypoints go through A.Resize
    together with THAT image (evaluator.py:209-219), i.e. they scale by its size, not by the DAIN frame's."""
    dain, _ = decode_resized_u8(dain_path, width, height, resize)
    gt = normalised_chw(decode_resized_u8(ref_img_path, width, height, resize)[0]) if is_key else None
    lm, conf = scaled_pose(pose_path, image_size(ref_img_path), width, height)
    pose = rasterise.frame_tables(lm, conf, height, width, thres1, thres2) if want_tables else (lm, conf)
    return dain, gt, pose


def save_png(u8, name, compress_level=None):
    """uint8 HWC -> file (utils/utils.py:139-142: Image.fromarray(...).save(name)); None = PIL's default zlib level (6), the
    bytes the reference writes."""
    from PIL import Image
    kw = {} if compress_level is None else {"compress_level": int(compress_level)}
    Image.fromarray(u8).save(name, **kw)
    return name


def warm():
    """First task of a fresh worker: pull in PIL's codecs and scipy's curve_fit before real work arrives."""
    from PIL import Image, PngImagePlugin, JpegImagePlugin      # noqa: F401
    from scipy.optimize import curve_fit                        # noqa: F401
    return True
