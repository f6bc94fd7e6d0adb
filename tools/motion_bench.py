#!/usr/bin/env python3
"""Stage-1 timing: clips/s and interpolated frames/s of the motion transformer on the
GPU for the clip lengths of the reference config, next to the CPU oracle on the host cores.

    python tools/motion_bench.py [--out profiles/rNN_motion.json]
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch                                                        # noqa: E402
import render_in_between_amd                                        # noqa: E402,F401
from render_in_between_amd.motion import MotionSpec, model, synth   # noqa: E402
from oracle import motion_ref                                       # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", type=str, default=None)
    ap.add_argument("--iters", type=int, default=50)
    ap.add_argument("--rounds", type=int, default=7, help="timed rounds (>= 5); the median is reported")
    args = ap.parse_args()
    spec = MotionSpec()
    sd = synth.make_state_dict(spec, 0)
    T = model.MotionTransformer(spec, device="cuda:0").eval()
    T.load_state_dict(sd)
    P = model.PositionEmbeddingSine1D(spec.pos_hidden_dim // 2)
    rows = []
    for N, n_key, rate in ((1, 9, 8), (1, 41, 8), (8, 41, 8), (32, 41, 8)):
        clips = [synth.make_clip(spec, n_key, rate, n) for n in range(N)]
        src = torch.stack([c[0] for c in clips]).cuda(); tgt = torch.stack([c[1] for c in clips]).cuda()
        sm = torch.stack([c[2] for c in clips]).cuda(); tm = torch.stack([c[3] for c in clips]).cuda()
        ps, pt = P(sm), P(tm)
        for _ in range(5):
            T(src, sm, ps, tgt, tm, pt, rate)
        torch.cuda.synchronize()
        # median of `rounds` timed rounds of `iters` forwards each (a single round has shown 3x outliers on a shared box)
        per_round = []
        for _ in range(args.rounds):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(args.iters):
                j, r = T(src, sm, ps, tgt, tm, pt, rate)
            e1.record(); torch.cuda.synchronize()
            per_round.append(e0.elapsed_time(e1) / args.iters)
        ms = sorted(per_round)[len(per_round) // 2]
        L = src.shape[-1]
        row = {"N": N, "L": L, "rate": rate, "gpu_ms": ms, "gpu_ms_min": min(per_round), "gpu_ms_max": max(per_round), "rounds": args.rounds,
               "gpu_frames_per_s": N * L / ms * 1e3, "launches": T._lib.ribm_num_launches(T._h)}
        if N == 1:
            npf = spec.pos_hidden_dim // 2
            c = [t.cpu() for t in (src, sm, tgt, tm)]
            t0 = time.time()
            reps = 3
            for _ in range(reps):
                oj, _ = motion_ref.transformer_forward(sd, spec.as_dict(), c[0], c[1], motion_ref.position_embedding_sine(c[1], npf), c[2], c[3],
                                                       motion_ref.position_embedding_sine(c[3], npf), rate)
            row["cpu_ms"] = (time.time() - t0) / reps * 1e3
            row["cpu_threads"] = torch.get_num_threads()
            row["max_abs_vs_oracle"] = float((j.cpu() - oj).abs().max())
        rows.append(row)
        print(row, flush=True)
    if args.out:
        with open(args.out, "w") as f:
            json.dump(rows, f, indent=1)


if __name__ == "__main__":
    main()
