#!/usr/bin/env python3
"""k_warp (the flow-grid bilinear warp the north star names) against its HBM roof.

    python3 tools/warp_bench.py --time [--out profiles/rNN_warp.json]          HIP-event timing, algorithmic GB/s
    rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d DIR -- python3 tools/warp_bench.py --run
    rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d DIR2 -- python3 tools/warp_bench.py --run
    python3 tools/warp_bench.py --report DIR DIR2 [--out ...]                   counter bytes per launch (FETCH x2: gfx950)

Algorithmic bytes per pixel: C channels of the frame read once + 2 flow components + C channels written = (2C + 2) * 4
(32 B at C = 3).  Shapes: 512x512 and 1024x1024, batch 4, smooth flows of +-3 px (every tap inside the staged window)
and of +-40 px (every pixel on the global-load path)."""
import argparse
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

SHAPES = [(4, 512, 512, 3.0), (4, 1024, 1024, 3.0), (4, 1024, 1024, 40.0)]
REPS = 20


def setup():
    import torch
    import render_in_between_amd as rib
    G = rib.Generator(rib.hsm_gen_config()).eval()        # rib_warp needs a handle, not weights
    cases = []
    for (B, H, W, amp) in SHAPES:
        g = torch.Generator().manual_seed(H + int(amp))
        img = (torch.rand(B, 3, H, W, generator=g) * 2 - 1).cuda()
        low = torch.randn(B, 2, H // 32, W // 32, generator=g)
        flow = (torch.nn.functional.interpolate(low, size=(H, W), mode="bilinear", align_corners=False).clamp(-1, 1) * amp).cuda()
        cases.append((B, H, W, amp, img, flow))
    return torch, G, cases


def run():
    torch, G, cases = setup()
    for (B, H, W, amp, img, flow) in cases:
        for _ in range(REPS):
            G.warp(img, flow)
    torch.cuda.synchronize()


def timeit(out):
    torch, G, cases = setup()
    rows = []
    for (B, H, W, amp, img, flow) in cases:
        for _ in range(5):
            G.warp(img, flow)
        torch.cuda.synchronize()
        ts = []
        for _ in range(7):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(REPS):
                o = G.warp(img, flow)
            e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) / REPS * 1e3)
        us = sorted(ts)[len(ts) // 2]
        ref = torch.nn.functional.grid_sample(
            img.cpu(), (torch.stack(torch.meshgrid(torch.linspace(-1, 1, H), torch.linspace(-1, 1, W), indexing="ij")[::-1], -1)[None]
                        + torch.stack([flow[:, 0].cpu() * 2 / (W - 1), flow[:, 1].cpu() * 2 / (H - 1)], -1)),
            mode="bilinear", padding_mode="border", align_corners=True)
        byt = B * H * W * (2 * 3 + 2) * 4
        rows.append({"B": B, "H": H, "W": W, "flow_amplitude_px": amp, "us_per_launch_median_of_7": us, "algorithmic_bytes": byt,
                     "algorithmic_GBps": byt / us / 1e3, "frac_of_8TBps": byt / us / 1e3 / 8000.0,
                     "max_abs_vs_grid_sample": float((o.cpu() - ref).abs().max())})
        print(rows[-1], flush=True)
    if out:
        with open(out, "w") as f:
            json.dump(rows, f, indent=1)


def report(dirs, out):
    per = {}
    for d, counter in zip(dirs, ("FETCH_SIZE", "WRITE_SIZE")):
        f = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)[0]
        disp = {}
        for r in csv.DictReader(open(f)):
            if "k_warp" in r["Kernel_Name"] and r["Counter_Name"] == counter:
                disp[int(r["Dispatch_Id"])] = disp.get(int(r["Dispatch_Id"]), 0.0) + float(r["Counter_Value"])
        ids = sorted(disp)
        assert len(ids) == REPS * len(SHAPES), (len(ids), counter)
        per[counter] = [sum(disp[k] for k in ids[i * REPS:(i + 1) * REPS]) / REPS * 1024.0 for i in range(len(SHAPES))]
    rows = []
    for i, (B, H, W, amp) in enumerate(SHAPES):
        byt = B * H * W * (2 * 3 + 2) * 4
        hbm = 2.0 * per["FETCH_SIZE"][i] + per["WRITE_SIZE"][i]
        rows.append({"B": B, "H": H, "W": W, "flow_amplitude_px": amp, "algorithmic_bytes": byt, "fetch_bytes_raw": per["FETCH_SIZE"][i],
                     "write_bytes": per["WRITE_SIZE"][i], "hbm_bytes_corrected (FETCH x2 + WRITE)": hbm, "traffic_over_algorithmic": hbm / byt})
        print(rows[-1])
    if out:
        with open(out, "w") as f:
            json.dump(rows, f, indent=1)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--run", action="store_true")
    ap.add_argument("--time", action="store_true")
    ap.add_argument("--report", nargs=2, default=None)
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    if a.run:
        run()
    elif a.time:
        timeit(a.out)
    else:
        report(a.report, a.out)
