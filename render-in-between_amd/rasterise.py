"""Host side of the GPU label rasteriser (device side: csrc/raster.hip.h
behind rib_rasterise).

What stays on the host is file parsing and per-limb scalar work:
  * read_json_keypoint   PGNR/utils/utils.py:12-60          OpenPose json -> 19x3 joints
  * valid_points         PGNR/utils/keypoint2img.py:115-131 thresholds, zero = joint off
  * stroke_table         keypoint2img.py:66-88,132-147      one fitted line per limb -> rib_stroke
  * peak_table           PGNR/datasets/HSM_auto_dataset.py:215-231   one-hot positions
  * gaussian_weights     the kernel scipy.ndimage.gaussian_filter builds (sigma, truncate 4)
Every pixel is produced on the GPU.  There is no CPU rasteriser in the product: the CPU
restatement lives in oracle/rasterise_ref.py and only the tests use it.

Line fit.  A limb joins two joints, so the reference's curve_fit(linear, ...) has an exact
answer, the line through them: a = dy/dx, b = y0 - a*x0.  curve_fit reaches it only to ~1e-12,
which matters in one situation: a sample a*x+b that lands (almost) on an integer, where the
truncation .astype(int) can go either way.  Those knife-edge limbs (integer-valued joints, in
practice only synthetic input) are detected and handed to scipy's curve_fit exactly as the
reference does, so the drawn pixels are identical in every case.
"""
from __future__ import annotations

import json
import warnings

import numpy as np

FOOT_IDX = (8, 9, 10, 11, 12, 13, 14, 15, 16)      # keypoint2img.py:121

POSE_EDGES = [[0, 1], [1, 8], [1, 2], [2, 3], [3, 4], [1, 5], [5, 6], [6, 7],
              [8, 9], [9, 10], [10, 11], [8, 12], [12, 13], [13, 14],
              [4, 18], [7, 17], [11, 16], [14, 15]]                       # keypoint2img.py:150-169 (p == 19)
POSE_COLORS = [[153, 0, 51], [153, 0, 0], [153, 51, 0], [153, 102, 0], [153, 153, 0],
               [102, 153, 0], [51, 153, 0], [0, 153, 0], [0, 153, 51], [0, 153, 102],
               [0, 153, 153], [0, 102, 153], [0, 51, 153], [0, 0, 153],
               [208, 208, 0], [0, 208, 0], [0, 208, 208], [0, 0, 208]]
STROKE_HALFWIDTH = 4                                                     # keypoint2img.py:145 (bw=4)

STROKE_DTYPE = np.dtype([("n", "<i4"), ("swap", "<i4"), ("start", "<f8"), ("step", "<f8"),
                         ("stop", "<f8"), ("a", "<f8"), ("b", "<f8")])   # == rib_stroke (include/rib.h)
assert STROKE_DTYPE.itemsize == 48


# ---- json ------------------------------------------------------------------------------------
def _mean_valid(pts, thres=0.0):
    valid = pts[:, 2] > thres
    return pts[valid].mean(axis=0, keepdims=True) if valid.sum() > 5 else np.zeros((1, 3))


def _largest_person(people, thres=0.1):
    best, best_area = -1, -1
    for i, person in enumerate(people):
        j = np.array(person["pose_keypoints_2d"], dtype=np.float64).reshape(-1, 3)[:15]
        valid = j[:, 2] > thres
        if valid.sum() < 4:
            continue
        area = (j[valid, 0].max() - j[valid, 0].min()) * (j[valid, 1].max() - j[valid, 1].min())
        if area > best_area:
            best, best_area = i, area
    return best


def read_json_keypoint(path):
    """OpenPose json -> (19, 3) [x, y, confidence]: body joints 0-14, toes 19 and 22, mean
    left / right hand (utils.py:12-60)."""
    with open(path) as f:
        d = json.load(f)
    people = d.get("people", [])
    idx = _largest_person(people) if people else -1
    if idx == -1:
        return np.zeros((19, 3))
    p = people[idx]
    body = np.array(p["pose_keypoints_2d"], dtype=np.float64).reshape(-1, 3)[list(range(15)) + [19, 22]]
    lh = _mean_valid(np.array(p["hand_left_keypoints_2d"], dtype=np.float64).reshape(-1, 3))
    rh = _mean_valid(np.array(p["hand_right_keypoints_2d"], dtype=np.float64).reshape(-1, 3))
    return np.concatenate([body, lh, rh], axis=0)


# ---- per-frame tables ------------------------------------------------------------------------
def valid_points(landmarks, conf, height, width, thres1=0.001, thres2=0.001):
    """(P, 2) joint positions, (0, 0) where the joint is off (keypoint2img.py:115-131)."""
    pts = np.zeros((len(landmarks), 2))
    for i, ((x, y), c) in enumerate(zip(landmarks, conf)):
        t = thres2 if i in FOOT_IDX else thres1
        if x >= 0 and y >= 0 and c > t and x < width and y < height:
            pts[i] = (x, y)
    return pts


def _knife_edge(vals):
    """True when truncating `vals` could flip under a ~1e-9 perturbation of the fitted line."""
    return bool(np.any(np.abs(vals - np.rint(vals)) < 1e-6 * np.maximum(1.0, np.abs(vals))))


def _lin(x, a, b):
    return a * x + b


def _stroke(x, y):
    """One limb -> (n, swap, start, step, stop, a, b); n = 0 when nothing is drawn (interpPoints,
    keypoint2img.py:66-88, for two points)."""
    x = np.asarray(x, np.float64); y = np.asarray(y, np.float64)
    swap = 0
    if abs(x[0] - x[1]) < abs(y[0] - y[1]):          # steep limb: fit x = a*y + b instead
        x, y, swap = y, x, 1
    fx, fy = x, y                                     # the fit sees the points in their given order
    if x[0] > x[1]:
        x, y = x[::-1], y[::-1]
    n = int(x[1] - x[0])
    if n <= 0:
        return (0, swap, 0.0, 0.0, 0.0, 0.0, 0.0)
    start, stop = int(x[0]), int(x[1])
    a = (fy[1] - fy[0]) / (fx[1] - fx[0])
    b = fy[0] - a * fx[0]
    cx = np.linspace(start, stop, n)
    if _knife_edge(a * cx + b):
        from scipy.optimize import curve_fit
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            (a, b), _ = curve_fit(_lin, fx, fy, maxfev=10000)
    step = (stop - start) / (n - 1) if n > 1 else 0.0     # np.linspace: step = delta / div
    return (n, swap, float(start), float(step), float(stop), float(a), float(b))


def stroke_table(pts, edges=POSE_EDGES):
    """(len(edges),) STROKE_DTYPE array for one frame (connect_keypoints, keypoint2img.py:132-147:
    a limb is drawn iff neither end has x == 0)."""
    out = np.zeros(len(edges), STROKE_DTYPE)
    for e, edge in enumerate(edges):
        if max(edge) >= len(pts):
            continue
        x, y = pts[edge, 0], pts[edge, 1]
        if 0 not in x:
            out[e] = _stroke(x, y)
    return out


def peak_table(landmarks, conf, height, width, thres=0.001):
    """(P, 2) int32 (x, y) of each joint's one-hot, (-1, -1) when the joint is off
    (HSM_auto_dataset.py:226-231)."""
    out = np.full((len(landmarks), 2), -1, np.int32)
    for i, ((x, y), c) in enumerate(zip(landmarks, conf)):
        if x >= 0 and y >= 0 and c > thres and x < width and y < height:
            out[i] = (int(x), int(y))
    return out


def gaussian_weights(sigma, truncate=4.0):
    """Half of scipy.ndimage's normalised gaussian kernel: w[d] is the weight of taps +-d,
    radius = int(truncate * sigma + 0.5)."""
    sigma = float(sigma)
    radius = int(truncate * sigma + 0.5)
    x = np.arange(-radius, radius + 1)
    phi = np.exp(-0.5 / (sigma * sigma) * x ** 2)
    phi = phi / phi.sum()
    return np.ascontiguousarray(phi[radius:], np.float64), radius


def frame_tables(landmarks, conf, height, width, thres1=0.001, thres2=0.001):
    pts = valid_points(landmarks, conf, height, width, thres1, thres2)
    return stroke_table(pts), peak_table(landmarks, conf, height, width, thres1)


def rasterise_labels(gen, frames, height, width, sigma=5, thres1=0.001, thres2=0.001):
    """frames: list of (landmarks, conf) already scaled to the model size.
    -> [T, 22, H, W] fp32 CUDA tensor, drawn by the GPU (gen.rasterise -> rib_rasterise)."""
    return rasterise_tables(gen, [frame_tables(lm, cf, height, width, thres1, thres2) for lm, cf in frames], height, width, sigma)


def rasterise_tables(gen, tabs, height, width, sigma=5):
    """Same, from per-frame host tables already built (frame_tables): the driver builds them in its
    decode workers so that the 0.8 ms/frame of host work overlaps the GPU."""
    strokes = np.stack([t[0] for t in tabs])
    peaks = np.stack([t[1] for t in tabs])
    w, radius = gaussian_weights(sigma)
    return gen.rasterise(strokes, peaks, w, radius, height, width)
