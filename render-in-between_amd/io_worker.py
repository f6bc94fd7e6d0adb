"""File-side work of the folder driver as plain functions of plain arguments (numpy / PIL only, no torch, no GPU): PNG /
JPEG decode + resize, OpenPose json -> rasteriser tables, PNG encode.  The Evaluator runs them either on threads of its
own process or - the default for the native path - in a pool of worker PROCESSES (forkserver): at several hundred
frames/s the PIL / scipy / json work of 32+ threads keeps the interpreter lock busy enough to starve the thread that
enqueues the GPU work (profiles/r04_driver.jsonl: the launch thread spent 0.1-0.3 s per clip waiting for the lock).

Reference semantics: PGNR/models/evaluator.py:205-235 (per-frame pre-load), PGNR/utils/utils.py:129-142 (save)."""
from __future__ import annotations

import os

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


# ---- shared-memory variants: the pixel payloads never go through a pipe -------------------------------------------------------
# A frame is 0.2-0.8 MB.  Pickled through multiprocessing's single result pipe it takes a dozen 64 KB reads in the parent's
# result thread, each of which has to win the interpreter lock back from the thread that enqueues the GPU work: the first
# process pool of this driver spent 2-4 s per clip there (profiles/r04_driver.jsonl).  So the parent hands out names of
# shared-memory blocks (Evaluator._ShmBlock: also page-locked for the GPU copies); workers decode INTO and encode OUT OF them,
# and only file names, offsets and the small rasteriser tables travel.
_SHM = {}


def _attach(name):
    shm = _SHM.get(name)
    if shm is None:
        from multiprocessing import shared_memory
        # a miss is the moment to forget blocks the parent has retired since (Evaluator._shm_trim unlinks them: the name is gone
        # from /dev/shm, but the pages stay allocated for as long as a process maps them)
        for gone in [n for n in _SHM if not os.path.exists("/dev/shm/" + n.lstrip("/"))]:
            _SHM.pop(gone).close()
        if len(_SHM) > 64:
            for old in _SHM.values():
                old.close()
            _SHM.clear()
        # (attaching registers the name with the resource tracker again; the workers share the parent's tracker process, where
        # that is a no-op on a set - the parent, which created the block, unlinks and unregisters it)
        shm = _SHM[name] = shared_memory.SharedMemory(name=name)
    return shm


def load_frame_shm(shm_name, offset, dain_path, ref_img_path, pose_path, is_key, want_tables, width, height, resize, thres1, thres2):
    """load_frame with the DAIN frame written to bytes [offset, offset + H*W*3) of a shared block (offset < 0: not wanted,
    the frame is a key frame that passes through); returns (None, key frame uint8 HWC or None, tables)."""
    if offset >= 0:
        dain, _ = decode_resized_u8(dain_path, width, height, resize)
        dst = np.ndarray((height, width, 3), np.uint8, buffer=_attach(shm_name).buf, offset=offset)
        dst[...] = dain
    gt = decode_resized_u8(ref_img_path, width, height, resize)[0] if is_key else None
    lm, conf = scaled_pose(pose_path, image_size(ref_img_path), width, height)
    pose = rasterise.frame_tables(lm, conf, height, width, thres1, thres2) if want_tables else (lm, conf)
    return None, gt, pose


def save_png_shm(shm_name, offset, height, width, name, compress_level=None):
    """save_png of the uint8 HWC frame at bytes [offset, ..) of a shared block."""
    return save_png(np.ndarray((height, width, 3), np.uint8, buffer=_attach(shm_name).buf, offset=offset), name, compress_level)


def warm():
    """Initialiser of a fresh worker process.  (1) One BLAS / OpenMP thread: numpy's OpenBLAS starts one thread per core of the
    host in EVERY process and they spin between calls - 32 workers x 256 threads on the GPU boxes made a 10 ms decode take 500 ms
    (tools/probes/io_pool_probe.py).  (2) Pull in PIL's codecs and scipy's curve_fit before real work arrives."""
    try:
        from threadpoolctl import threadpool_limits
        threadpool_limits(1)
    except Exception:                                           # noqa: BLE001 (no threadpoolctl: the environment variables below apply)
        pass
    from PIL import Image, PngImagePlugin, JpegImagePlugin      # noqa: F401
    from scipy.optimize import curve_fit                        # noqa: F401
    return True
