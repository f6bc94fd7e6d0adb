#!/usr/bin/env python3
"""Label rasterisation rate: GPU (rib_rasterise, whole clip per call) vs the CPU oracle per frame.

    python tools/raster_bench.py [--size 512] [--frames 256] [--cpu-frames 3]
"""
import argparse, json, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import render_in_between_amd as rib
from render_in_between_amd import rasterise as R


def person(rng, H, W):
    c = np.array([W * rng.uniform(0.35, 0.65), H * rng.uniform(0.4, 0.6)])
    s = min(H, W) / 3.2
    base = np.array([[0, -1.0], [0, -0.7], [-0.3, -0.7], [-0.45, -0.3], [-0.5, 0.1], [0.3, -0.7], [0.45, -0.3],
                     [0.5, 0.1], [0, 0.0], [-0.15, 0.0], [-0.2, 0.5], [-0.2, 1.0], [0.15, 0.0], [0.2, 0.5],
                     [0.2, 1.0], [0.3, 1.1], [-0.3, 1.1], [0.55, 0.15], [-0.55, 0.15]])
    xy = np.round(base * s + rng.normal(0, 0.03 * s, base.shape) + c, 3)
    return [tuple(v) for v in xy], list(rng.uniform(0.3, 0.95, 19))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", type=int, default=512)
    ap.add_argument("--frames", type=int, default=256)
    ap.add_argument("--cpu-frames", type=int, default=3)
    ap.add_argument("--reps", type=int, default=5)
    a = ap.parse_args()
    H = W = a.size
    rng = np.random.default_rng(0)
    frames = [person(rng, H, W) for _ in range(a.frames)]
    G = rib.Generator(rib.hsm_gen_config(num_filters=16, max_num_filters=64, mask=dict(num_filters=32, max_num_filters=64),
                                         embed=dict(num_filters=32, max_num_filters=64)))
    t0 = time.perf_counter()
    tabs = [R.frame_tables(lm, cf, H, W) for lm, cf in frames]
    host_ms = (time.perf_counter() - t0) * 1e3 / a.frames
    strokes = np.stack([t[0] for t in tabs]); peaks = np.stack([t[1] for t in tabs])
    w, r = R.gaussian_weights(5)
    out = {}
    for T in sorted({1, 32, a.frames}):
        G.rasterise(strokes[:T], peaks[:T], w, r, H, W)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.reps):
            lab = G.rasterise(strokes[:T], peaks[:T], w, r, H, W)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) * 1e3 / a.reps
        out["gpu_T%d" % T] = {"ms_per_call": ms, "frames_per_s": T / ms * 1e3}
    res = {"size": a.size, "host_tables_ms_per_frame": host_ms, **out}
    if a.cpu_frames:
        from oracle import rasterise_ref as O               # the CPU side of the comparison only
        t0 = time.perf_counter()
        for lm, cf in frames[:a.cpu_frames]:
            sk = O.skeleton_image(lm, cf, H, W); pm = O.pose_map(lm, cf, H, W)
        cpu_ms = (time.perf_counter() - t0) * 1e3 / a.cpu_frames
        res["cpu_oracle_ms_per_frame"] = cpu_ms
        got = lab[a.cpu_frames - 1].cpu().numpy() if a.cpu_frames <= a.frames else None
        want_sk = ((sk.astype(np.float32) / 255.0 - 0.5) / 0.5).transpose(2, 0, 1)
        res["bit_exact_vs_oracle"] = bool(np.array_equal(got[:3], want_sk) and np.array_equal(got[3:], pm))
    print(json.dumps(res))


if __name__ == "__main__":
    main()
