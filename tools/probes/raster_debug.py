"""Per-limb comparison of the GPU painter with the CPU oracle (debug aid for rib_rasterise)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import render_in_between_amd as rib
from render_in_between_amd import rasterise as R, synth
from oracle import rasterise_ref as O

H, W = 96, 160
rng = np.random.default_rng(11)
cfg = rib.hsm_gen_config(num_filters=16, max_num_filters=64, mask=dict(num_filters=32, max_num_filters=64), embed=dict(num_filters=32, max_num_filters=64))
G = rib.Generator(cfg)
w, r = R.gaussian_weights(5)
bad = 0
for it in range(300):
    x = np.round(rng.uniform(1, W - 1, 2)); y = np.round(rng.uniform(1, H - 1, 2))
    if it % 3 == 0:
        x = np.round(rng.uniform(1, W - 1, 2), 3); y = np.round(rng.uniform(1, H - 1, 2), 3)
    pts = np.zeros((2, 2)); pts[:, 0] = x; pts[:, 1] = y
    st = R.stroke_table(pts, edges=[[0, 1]])
    peaks = np.full((1, 19, 2), -1, np.int32)
    got = G.rasterise(st[None], peaks, w, r, H, W, colors=[[153, 51, 0]]).cpu().numpy()[0, :3]
    img = np.zeros((H, W, 3), np.uint8)
    cx, cy = O._interp_points(x, y)
    O._draw_edge(img, cx, cy, 4, [153, 51, 0])
    want = ((img.astype(np.float32) / 255.0 - 0.5) / 0.5).transpose(2, 0, 1)
    nd = int((got != want).any(0).sum())
    if nd:
        bad += 1
        ys, xs = np.nonzero((got != want).any(0))
        A = (cx[0], cy[0]); B = (cx[-1], cy[-1])
        inside = all(12 <= p[0] <= W - 12 and 12 <= p[1] <= H - 12 for p in (A, B))
        apart = abs(A[0] - B[0]) >= 24 or abs(A[1] - B[1]) >= 24
        print(it, "x", x, "y", y, "n", st[0]["n"], "swap", st[0]["swap"], "diff px", nd, "bbox", xs.min(), xs.max(), ys.min(), ys.max(), "A", A, "B", B, "inside", inside, "apart", apart)
print("bad limbs", bad)
