#!/usr/bin/env python3
"""Join a rocprofv3 --kernel-trace CSV with the launch plan: per-op device time.

    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof -- python3 tools/prof_ops.py --run
    python3 tools/prof_ops.py --report gpurun_out/prof > gpurun_out/prof_ops.txt

--run executes W warm-up and K timed forward+blend steps - bench.py's default step - and nothing else
launches kernels, so dispatches map onto plan ops by position.

Per launch the report prices a floor  min_us = max(algorithmic FLOPs / dense MFMA peak of the dtype, algorithmic bytes /
6.3 TB/s achievable HBM)  (FLOPs: SURVEY 8(d)'s nine-tap count of the layer, so a Winograd or phase-decomposed launch can
exceed 1.0; bytes: every operand read once, every result written once - rib_debug_launch_info), the fraction min_us / us
and the time lost to it, us - min_us; a second table lists the launches by lost time, largest first.
"""
import argparse
import csv
import ctypes as C
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def plan_names(B, H, W, dtype="f32"):
    import render_in_between_amd as rib
    from render_in_between_amd import _native
    lib = _native.lib()
    spec = rib.GenSpec.from_cfg(rib.hsm_gen_config())
    c = _native.RibConfig(**{n: getattr(spec, n) for n, _ in _native.RibConfig._fields_})
    h = C.c_void_p()
    assert lib.rib_create(C.byref(c), -1, C.byref(h)) == 0
    from render_in_between_amd import tuning
    assert lib.rib_set_compute_dtype(h, {"f32": 0, "bf16": 1, "f16": 3}[dtype]) == 0
    tuning.apply(lib, h, tuning.load(dtype=dtype), B, H, W, dtype=dtype)      # the same pinned choices Generator applies
    buf = C.create_string_buffer(512)
    out = []
    for i in range(lib.rib_num_launches(h, B, H, W)):
        lib.rib_debug_launch_info(h, B, H, W, i, buf, 512)
        name, kclass, grid, tile, flops, nbytes = buf.value.decode().split("|")
        out.append({"name": name, "class": int(kclass), "grid": grid, "tile": tile, "flops": float(flops), "bytes": float(nbytes)})
    return out


def run(args):
    import torch
    import render_in_between_amd as rib
    from render_in_between_amd import synth
    cfg = rib.hsm_gen_config()
    spec = rib.GenSpec.from_cfg(cfg)
    G = rib.Generator(cfg, compute_dtype=args.dtype, products=args.products).eval()
    G.load_state_dict(synth.make_state_dict(spec, 0))
    label, fake, prev = [t.cuda() for t in synth.make_inputs(spec, args.batch, args.height or args.size, args.width or args.size, 0)]
    torch.cuda.synchronize()
    for _ in range(args.warmup + args.steps):
        G.forward_blend(label, None, fake, prev)           # bench.py's step: the mask head writes the fused frame
    torch.cuda.synchronize()


def report(args):
    files = glob.glob(os.path.join(args.report, "**", "*kernel_trace.csv"), recursive=True)
    assert files, "no kernel_trace.csv under " + args.report
    rows = []
    with open(files[0]) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    rows.sort()
    ops = plan_names(args.batch, args.height or args.size, args.width or args.size, args.dtype)
    rows = [r for r in rows if r[2].startswith(("void rib::", "rib::"))]
    n = len(ops)
    total_steps = len(rows) // n
    assert total_steps >= 1, (len(rows), n)
    rows = rows[-n * min(args.steps, total_steps):]
    steps = len(rows) // n
    agg = [0.0] * n
    for i, r in enumerate(rows):
        agg[i % n] += (r[1] - r[0]) / 1e3 / steps          # us
    span = sum((rows[(s + 1) * n - 1][1] - rows[s * n][0]) for s in range(steps)) / steps / 1e3
    print("# %d steps, %d launches/step, sum of kernel time %.1f us, first-start to last-end %.1f us/step"
          % (steps, n, sum(agg), span))
    cls = {}
    for o, t in zip(ops, agg):
        cls.setdefault(o["class"], [0, 0.0]); cls[o["class"]][0] += 1; cls[o["class"]][1] += t
    names = ("igemm", "spade", "stats", "pool", "eltwise", "pack", "conv_aux")
    for k in sorted(cls):
        print("# class %-8s launches %3d  %.1f us" % (names[k], cls[k][0], cls[k][1]))
    peak_tf = 157.3 if args.dtype == "f32" else 2500.0      # dense MFMA peak of the dtype (MI355X_MICROARCH.md)
    hbm = 6.3e12                                            # achievable HBM rate (float4 copy: 6.29 TB/s)
    for o, t in zip(ops, agg):
        t_f = o["flops"] / (peak_tf * 1e12) * 1e6
        t_b = o["bytes"] / hbm * 1e6
        o["min_us"] = max(t_f, t_b)
        o["bound"] = "mfma" if t_f >= t_b else "hbm"
        o["frac"] = o["min_us"] / t if t > 0 else 0.0
        o["lost_us"] = t - o["min_us"]
    tot_min = sum(o["min_us"] for o in ops)
    print("# floor of the step: sum of per-launch min_us %.1f us = %.3f of the kernel time; lost %.1f us" % (tot_min, tot_min / sum(agg), sum(agg) - tot_min))
    print("%-52s %9s %8s %8s %6s %5s %8s  %-14s %s" % ("op", "us", "TFLOP/s", "min_us", "bound", "frac", "lost_us", "grid", "tile"))
    for o, t in zip(ops, agg):
        tf = o["flops"] / (t * 1e-6) / 1e12 if t > 0 and o["flops"] else 0.0
        print("%-52s %9.1f %8.1f %8.1f %6s %5.2f %8.1f  %-14s %s" % (o["name"], t, tf, o["min_us"], o["bound"], o["frac"], o["lost_us"], o["grid"], o["tile"]))
    print("\n# launches by time lost against their own floor (us - min_us), largest first")
    print("%-52s %9s %8s %5s %8s %7s" % ("op", "us", "min_us", "frac", "lost_us", "cum_us"))
    cum = 0.0
    for o, t in sorted(zip(ops, agg), key=lambda x: -x[0]["lost_us"]):
        cum += o["lost_us"]
        print("%-52s %9.1f %8.1f %5.2f %8.1f %7.0f" % (o["name"], t, o["min_us"], o["frac"], o["lost_us"], cum))
    if args.json:
        with open(args.json, "w") as f:
            json.dump({"steps": steps, "sum_us": sum(agg), "span_us": span, "floor_us": tot_min,
                       "ops": [dict(o, us=t) for o, t in zip(ops, agg)]}, f, indent=0)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--run", action="store_true")
    ap.add_argument("--report", type=str, default=None)
    ap.add_argument("--json", type=str, default=None)
    ap.add_argument("--size", type=int, default=512)
    ap.add_argument("--height", type=int, default=0)
    ap.add_argument("--width", type=int, default=0)
    ap.add_argument("--batch", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--dtype", choices=("f32", "bf16", "f16"), default="f32")
    ap.add_argument("--products", choices=("f32", "bf16x3"), default="f32")
    a = ap.parse_args()
    if a.run:
        run(a)
    else:
        report(a)
