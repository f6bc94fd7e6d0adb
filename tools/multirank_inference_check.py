#!/usr/bin/env python3
"""The product's multi-GPU path, end to end, on whatever GPUs the box has: writes a synthetic example folder
(inputs / DAIN / Predict_motion, two clips) and a seed-defined checkpoint, runs `inference.py` once with one rank and once
with N ranks (the CLI starts its own ranks; on a 1-GPU box set RIB_BENCH_DEVICE=0 RIB_DIST_BACKEND=gloo so that all ranks
share the device - at most 5 there: the pool's process guard allows 6 processes on the card and the launching process counts), with the CLI's DEFAULT settings (batched segments, batch-invariant
plans), and compares the written PNGs byte for byte: an N-rank run must write exactly the files a 1-rank run writes.  This process never touches the GPU: the CLIs run as
child processes.

    RIB_BENCH_DEVICE=0 RIB_DIST_BACKEND=gloo python3 tools/multirank_inference_check.py --gpus 2 [--dtype f32]
"""
import argparse, json, os, shutil, subprocess, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import yaml


def write_clip(root, clip, n_key, rate, H, W, seed):
    from PIL import Image
    rng = np.random.default_rng(seed)
    n = (n_key - 1) * rate + 1
    for d in ("inputs", "DAIN", "Predict_motion"):
        os.makedirs(os.path.join(root, d, clip))
    for k in range(n_key):
        Image.fromarray(rng.integers(0, 255, (H, W, 3), dtype=np.uint8)).save(os.path.join(root, "inputs", clip, "%04d.png" % k))
    for i in range(n):
        Image.fromarray(rng.integers(0, 255, (H, W, 3), dtype=np.uint8)).save(os.path.join(root, "DAIN", clip, "f%03d.png" % i))
        body = []
        for j in range(25):
            body += [float(rng.uniform(4, W - 4)), float(rng.uniform(4, H - 4)), 0.9]
        hand = [10.0, 10.0, 0.9] * 21
        with open(os.path.join(root, "Predict_motion", clip, "f%03d_keypoints.json" % i), "w") as f:
            json.dump({"people": [{"pose_keypoints_2d": body, "hand_left_keypoints_2d": hand, "hand_right_keypoints_2d": hand}]}, f)
    return n


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=2)
    ap.add_argument("--dtype", default="f32")
    ap.add_argument("--size", type=int, nargs=2, default=[128, 192])
    ap.add_argument("--keys", type=int, nargs="+", default=[4, 3], help="key frames per clip (one clip per entry; frame rates 4, 2, 4, ...): "
                                                                        "--keys 9 6 5 = 8 + 5 + 4 segments, enough for six ranks")
    ap.add_argument("--batch", type=int, default=0, help="0 (default): the driver's own batching by frame size; N: chains of N segments")
    ap.add_argument("--no-reproducible", action="store_true", help="pass --no-reproducible to the CLI: every group size runs its own measured table; "
                                                                   "frames may then differ by one uint8 step between world sizes")
    a = ap.parse_args()
    import render_in_between_amd as rib
    from render_in_between_amd import synth
    tmp = tempfile.mkdtemp(prefix="rib_mr_")
    H, W = a.size
    n = sum(write_clip(tmp, "clip%c" % (65 + i), k, 4 if i % 2 == 0 else 2, H, W, 1 + i) for i, k in enumerate(a.keys))      # default: 3 + 2 segments, 13 + 5 frames
    spec = rib.GenSpec.from_cfg(rib.hsm_gen_config())
    ck = os.path.join(tmp, "netG.pth")
    torch.save(synth.make_state_dict(spec, 3), ck)
    cfg = yaml.load(open(os.path.join(ROOT, "render-in-between_amd", "configs", "HSM.yaml")), Loader=yaml.FullLoader)
    cfg["model_pretrain_G"] = ck; cfg["model_height"] = H; cfg["model_width"] = W
    cpath = os.path.join(tmp, "cfg.yaml")
    yaml.dump(cfg, open(cpath, "w"))
    cli = os.path.join(ROOT, "render-in-between_amd", "inference.py")
    res = {}
    for g in (1, a.gpus):
        out = os.path.join(tmp, "out%d" % g)
        t0 = time.time()
        r = subprocess.run([sys.executable, cli, "--config", cpath, "--input-dir", tmp, "--save-dir", out, "--gpus", str(g), "--dtype", a.dtype, "--batch", str(a.batch)] + (["--no-reproducible"] if a.no_reproducible else []),
                           capture_output=True, text=True)
        if r.returncode != 0:
            print(r.stdout[-2000:], r.stderr[-4000:]); raise SystemExit("inference.py --gpus %d failed (rc %d)" % (g, r.returncode))
        res[g] = (out, time.time() - t0, [l for l in r.stdout.splitlines() if "rank" in l or "broadcast" in l])
    files = sorted(os.path.relpath(os.path.join(d, f), res[1][0]) for d, _, fs in os.walk(res[1][0]) for f in fs)
    assert len(files) == n, (len(files), n)
    from PIL import Image
    same, worst = 0, 0
    for f in files:
        same += open(os.path.join(res[1][0], f), "rb").read() == open(os.path.join(res[a.gpus][0], f), "rb").read()
        worst = max(worst, int(np.abs(np.asarray(Image.open(os.path.join(res[1][0], f))).astype(int) - np.asarray(Image.open(os.path.join(res[a.gpus][0], f))).astype(int)).max()))
    print(json.dumps({"frames": n, "ranks": a.gpus, "dtype": a.dtype, "size": [H, W], "batch": a.batch, "reproducible": not a.no_reproducible, "png_files_byte_identical": same,
                      "all_identical": same == n, "max_abs_pixel_difference": worst, "seconds_1_rank": round(res[1][1], 2), "seconds_%d_ranks" % a.gpus: round(res[a.gpus][1], 2),
                      "rank_lines": res[a.gpus][2], "one_rank_line": res[1][2]}))
    shutil.rmtree(tmp)
    if (not a.no_reproducible and same != n) or worst > 1:      # the default policy promises the same bytes at every world size
        raise SystemExit(1)


if __name__ == "__main__":
    main()
