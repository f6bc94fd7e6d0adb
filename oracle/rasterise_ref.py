"""TEST INFRASTRUCTURE — CPU restatement of the reference's input rasterisation
(SURVEY 8 row f-2); the product path is the GPU rasteriser behind rib_rasterise
(render-in-between_amd/rasterise.py + csrc/raster.hip.h).  Only tests/, smoke() and
bench.py's cpu_baseline leg may import this module.

Restates, from reading the reference:
  * read_json_keypoint            PGNR/utils/utils.py:12-60      OpenPose json -> 19x3 joints
  * _generate_pose_map (test)     PGNR/datasets/HSM_auto_dataset.py:205-236
        one-hot at (int(y), int(x)) -> scipy gaussian_filter(sigma) -> / max
  * _generate_skeleton            PGNR/datasets/HSM_auto_dataset.py:238-251 and
    PGNR/utils/keypoint2img.py:30-173   coloured limb curves (line / quadratic curve_fit, bw=4)

Parity status: PINNED.  tests/golden/make_golden_raster.py runs the reference's own
functions in the build container (keypoint2img.py and utils.py imported as they are;
the two dataset METHODS compiled out of HSM_auto_dataset.py's syntax tree, because
the module itself needs cv2 / albumentations / h5py) and commits their outputs as
tests/golden/raster_*.npz; this restatement reproduces every one bit for bit
(tests/test_oracle_golden.py).
"""
from __future__ import annotations

import json
import warnings

import numpy as np
from scipy import ndimage
from scipy.optimize import curve_fit

FOOT_IDX = (8, 9, 10, 11, 12, 13, 14, 15, 16)      # keypoint2img.py:121

POSE_EDGES = [[0, 1], [1, 8], [1, 2], [2, 3], [3, 4], [1, 5], [5, 6], [6, 7],
              [8, 9], [9, 10], [10, 11], [8, 12], [12, 13], [13, 14],
              [4, 18], [7, 17], [11, 16], [14, 15]]                       # keypoint2img.py:150-169 (p == 19)
POSE_COLORS = [[153, 0, 51], [153, 0, 0], [153, 51, 0], [153, 102, 0], [153, 153, 0],
               [102, 153, 0], [51, 153, 0], [0, 153, 0], [0, 153, 51], [0, 153, 102],
               [0, 153, 153], [0, 102, 153], [0, 51, 153], [0, 0, 153],
               [208, 208, 0], [0, 208, 0], [0, 208, 208], [0, 0, 208]]


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


def pose_map(landmarks, conf, height, width, sigma=5, thres=0.001):
    """(19, H, W) float32 heat-maps in [0, 1], test-phase branch of _generate_pose_map."""
    maps = np.zeros((len(landmarks), height, width), np.float64)
    for i, ((x, y), c) in enumerate(zip(landmarks, conf)):
        if x >= 0 and y >= 0 and c > thres and x < width and y < height:
            m = np.zeros((height, width))
            m[int(y), int(x)] = 1
            m = ndimage.gaussian_filter(m, sigma=sigma)
            maps[i] = m / m.max()
    return maps.astype(np.float32)


def _quad(x, a, b, c):
    return a * x ** 2 + b * x + c


def _lin(x, a, b):
    return a * x + b


def _interp_points(x, y):
    """keypoint2img.py:66-88."""
    if abs(x[:-1] - x[1:]).max() < abs(y[:-1] - y[1:]).max():
        cy, cx = _interp_points(y, x)
        return (None, None) if cy is None else (cx, cy)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        if len(x) < 3:
            popt, _ = curve_fit(_lin, x, y, maxfev=10000)
        else:
            popt, _ = curve_fit(_quad, x, y)
            if abs(popt[0]) > 1:
                return None, None
    if x[0] > x[-1]:
        x = np.array(list(reversed(x))); y = np.array(list(reversed(y)))
    cx = np.linspace(int(x[0]), int(x[-1]), int(x[-1] - x[0]))
    cy = _lin(cx, *popt) if len(x) < 3 else _quad(cx, *popt)
    return cx.astype(int), cy.astype(int)


def _set_color(im, yy, xx, color):
    """keypoint2img.py:36-45: paint, averaging with what is already there."""
    if (im[yy, xx] == 0).all():
        im[yy, xx] = color
    else:
        im[yy, xx] = ((im[yy, xx].astype(float) + np.array(color, dtype=float)) / 2).astype(np.uint8)


def _draw_edge(im, x, y, bw, color):
    """keypoint2img.py:47-64 with draw_end_points=True."""
    if x is None or not x.size:
        return
    h, w = im.shape[:2]
    for i in range(-bw, bw):
        for j in range(-bw, bw):
            _set_color(im, np.clip(y + i, 0, h - 1), np.clip(x + j, 0, w - 1), color)
    for i in range(-bw * 3, bw * 3):
        for j in range(-bw * 3, bw * 3):
            if i * i + j * j < 4 * bw * bw:
                _set_color(im, np.clip(np.array([y[0], y[-1]]) + i, 0, h - 1),
                           np.clip(np.array([x[0], x[-1]]) + j, 0, w - 1), color)


def skeleton_image(landmarks, conf, height, width, thres1=0.001, thres2=0.001):
    """(H, W, 3) uint8 limb drawing (_generate_skeleton, test phase: no random drops)."""
    pts = np.zeros((len(landmarks), 2))
    for i, ((x, y), c) in enumerate(zip(landmarks, conf)):
        t = thres2 if i in FOOT_IDX else thres1
        if x >= 0 and y >= 0 and c > t and x < width and y < height:      # keypoint2img.py:115-131
            pts[i] = (x, y)
    img = np.zeros((height, width, 3), np.uint8)
    for edge, color in zip(POSE_EDGES, POSE_COLORS):
        if max(edge) >= len(pts):
            continue
        x, y = pts[edge, 0], pts[edge, 1]
        if 0 not in x:                                                      # keypoint2img.py:143
            cx, cy = _interp_points(x, y)
            _draw_edge(img, cx, cy, 4, color)
    return img
