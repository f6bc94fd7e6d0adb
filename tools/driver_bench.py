#!/usr/bin/env python3
"""End-to-end rate of the folder driver (json + PNG in, PNG out) on a synthetic clip.

    python tools/driver_bench.py [--size 512 | --height 320 --width 480] [--keys 5] [--rate 32] [--lanes 2] [--batch B] [--chunk 4]

Writes a clip in the reference's directory layout (inputs/ DAIN/ Predict_motion/), runs
Evaluator.evaluate_from_folder twice (the first run also builds launch plans) and prints the
phase times of the second: load (decode + json), rasterise (GPU), generate (GPU chains + quantise +
one D2H copy), save (PNG encode).
"""
import argparse, json, os, sys, tempfile, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import render_in_between_amd as rib
from render_in_between_amd import evaluator as ev, synth
from tools.raster_bench import person


def write_clip(root, n_key, rate, H, W):
    from PIL import Image
    rng = np.random.default_rng(0)
    n = (n_key - 1) * rate + 1
    for d in ("inputs", "DAIN", "Predict_motion"):
        os.makedirs(os.path.join(root, d, "clip"))
    spec = rib.GenSpec.from_cfg(rib.hsm_gen_config())
    def img(seed):
        a = np.asarray(synth.smooth_image(spec, 1, H, W, seed))[0]
        return ((a * 0.5 + 0.5).clip(0, 1) * 255).astype(np.uint8).transpose(1, 2, 0)
    for k in range(n_key):
        Image.fromarray(img(k)).save(os.path.join(root, "inputs", "clip", "%04d.png" % k))
    for i in range(n):
        Image.fromarray(img(100 + i)).save(os.path.join(root, "DAIN", "clip", "f%04d.png" % i))
        lm, conf = person(rng, H, W)
        body = np.zeros((25, 3)); idx = list(range(15)) + [19, 22]
        for j, k in enumerate(idx):
            body[k] = (lm[j][0], lm[j][1], conf[j])
        hand = lambda c: [v for _ in range(21) for v in (c[0] + float(rng.normal(0, 3)), c[1] + float(rng.normal(0, 3)), 0.8)]
        doc = {"people": [{"pose_keypoints_2d": [float(v) for v in body.reshape(-1)],
                           "hand_left_keypoints_2d": hand(lm[17]), "hand_right_keypoints_2d": hand(lm[18])}]}
        with open(os.path.join(root, "Predict_motion", "clip", "f%04d_keypoints.json" % i), "w") as f:
            json.dump(doc, f)
    return n


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", type=int, default=512)
    ap.add_argument("--keys", type=int, default=3)
    ap.add_argument("--rate", type=int, default=32)
    ap.add_argument("--height", type=int, default=0)
    ap.add_argument("--width", type=int, default=0)
    ap.add_argument("--lanes", type=int, default=2)
    ap.add_argument("--batch", type=int, default=0, help="segments per chain (0: the Evaluator's default for the frame size)")
    ap.add_argument("--chunk", type=int, default=4)
    ap.add_argument("--io-threads", type=int, default=0)
    ap.add_argument("--compress", type=int, default=-1, help="PNG compress level (-1: PIL's default, 6, as the reference)")
    ap.add_argument("--io-mode", default="process", choices=("process", "thread"))
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--dtype", default="f32")
    a = ap.parse_args()
    H, W = (a.height or a.size), (a.width or a.size)
    cfg = rib.AttrDict(gen=rib.hsm_gen_config(), model_height=H, model_width=W, gauss_sigma=5, skeleton_thres=0.001, foot_thres=0.001)
    spec = rib.GenSpec.from_cfg(cfg.gen)
    G = rib.Generator(cfg.gen, compute_dtype=a.dtype).eval()
    G.load_state_dict(synth.make_state_dict(spec, 0, power_iters=3))
    with tempfile.TemporaryDirectory() as root:
        n = write_clip(root, a.keys, a.rate, H, W)
        E = ev.Evaluator(cfg, lanes=a.lanes, batch=a.batch or None, chunk=a.chunk, io_threads=a.io_threads or None,
                         png_compress_level=None if a.compress < 0 else a.compress, io_mode=a.io_mode)
        dirs = [os.path.join(root, d) for d in ("inputs", "DAIN", "Predict_motion")]
        walls = []
        for rep in range(1 + a.reps):           # the first run also builds launch plans and pools: not counted
            t0 = time.perf_counter()
            out = E.evaluate_from_folder(G, *dirs, os.path.join(root, "out%d" % rep))
            torch.cuda.synchronize()
            walls.append(time.perf_counter() - t0)
        tm = dict(E.timings)
    gen = n - a.keys
    wall = sorted(walls[1:])[len(walls[1:]) // 2]
    print(json.dumps({"height": H, "width": W, "dtype": a.dtype, "frames": n, "generated": gen, "lanes": a.lanes,
                      "batch": a.batch or E.default_batch(), "chunk": a.chunk, "io_threads": E.io_threads, "io_mode": a.io_mode,
                      "cpus": len(os.sched_getaffinity(0)), "cpu_budget": ev.cpu_budget(), "png_compress_level": a.compress,
                      "wall_s": wall, "wall_s_runs": [round(w, 4) for w in walls[1:]], "frames_per_s_end_to_end": n / wall,
                      "phase_s_last_run": {k: round(v, 4) for k, v in tm.items() if k not in ("frames", "units", "timeline", "peak_units_in_flight")},
                      "peak_units_in_flight": tm.get("peak_units_in_flight"),
                      "unit_timeline_s [decoded, enqueued, on host, written]": tm.get("timeline"),
                      "pipeline_units_last_run": tm.get("units")}))


if __name__ == "__main__":
    main()
