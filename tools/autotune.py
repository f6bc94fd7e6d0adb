#!/usr/bin/env python3
"""Measure, per convolution / SPADE launch of the plan, every compatible (tile variant, split-K)
choice on the GPU in isolation and pin the fastest.  Results are merged into
render-in-between_amd/tuning_gfx950.json, which Generator applies at plan-build time.

    python3 tools/autotune.py --size 512 --batch 1 [--iters 20]
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch                                                   # noqa: E402
import render_in_between_amd as rib                            # noqa: E402
from render_in_between_amd import _native, synth, tuning      # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", type=int, nargs="+", default=[512])
    ap.add_argument("--width", type=int, default=0)
    ap.add_argument("--batch", type=int, default=1)
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--dtype", choices=("f32", "bf16", "f16"), default="f32", help="precision mode whose kernels are measured")
    ap.add_argument("--out", type=str, default=None)
    ap.add_argument("--report", type=str, default=None)
    ap.add_argument("--only", type=str, default=None,
                    help="re-tune only the launches whose name contains this substring; other entries of the shape are kept")
    args = ap.parse_args()
    if args.out is None:
        args.out = tuning.TUNING_PATHS[args.dtype]
    lib = _native.lib()
    cfg = rib.hsm_gen_config()
    spec = rib.GenSpec.from_cfg(cfg)
    G = rib.Generator(cfg, use_tuning=False, compute_dtype=args.dtype).eval()
    G.load_state_dict(synth.make_state_dict(spec, 0))
    h = G._h
    nvar = lib.rib_num_variants()
    geoms = []
    g10 = (C.c_int * 12)()
    for i in range(nvar):
        prec = lib.rib_variant_info(i, g10)
        geoms.append(list(g10) if prec == tuning.PREC[args.dtype] else None)      # only the kernels of this precision mode
    table = tuning.load(args.out)
    report = []
    for size in args.size:
        B, H, W = args.batch, size, (args.width or size)
        label, fake, prev = [t.cuda() for t in synth.make_inputs(spec, B, H, W, 0)]
        img = torch.empty((B, 3, H, W), device="cuda"); mask = torch.empty((B, 1, H, W), device="cuda")
        ws = torch.empty(int(lib.rib_workspace_bytes(h, B, H, W) * 2.5), dtype=torch.uint8, device="cuda")
        st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        args_fwd = [C.c_void_p(t.data_ptr()) for t in (label, fake, prev, img, mask)]
        _native.check(h, lib.rib_forward(h, B, H, W, *args_fwd, C.c_void_p(ws.data_ptr()), ws.numel(), st))
        torch.cuda.synchronize()
        buf = C.create_string_buffer(512)
        ops = []
        for i in range(lib.rib_num_launches(h, B, H, W)):
            lib.rib_debug_launch_info(h, B, H, W, i, buf, 512)
            name, kclass, grid, tile, flops, _ = buf.value.decode().split("|")
            if tile.startswith("lowc"):
                continue            # k_conv_lowc has one implementation per layer shape: nothing to choose
            if int(kclass) in (0, 1) and (args.only is None or args.only in name):
                ops.append((name, int(kclass), float(flops), tile))
        usec = C.c_double()
        chosen = {}
        t_start = time.time()
        total_before = total_after = 0.0
        for name, kclass, flops, tile0 in ops:
            def timeit():
                n = lib.rib_workspace_bytes(h, B, H, W)
                if n == 0 or n > ws.numel():
                    return None
                rc = lib.rib_time_op(h, B, H, W, name.encode(), *args_fwd, C.c_void_p(ws.data_ptr()), ws.numel(),
                                     args.iters, st, C.byref(usec))
                return usec.value if rc == 0 else None
            lib.rib_set_choice(h, B, H, W, name.encode(), -1, 1)
            base = timeit()
            best = (base, None, None)
            results = []
            for vi, g in enumerate(geoms):
                if g is None:
                    continue
                if kclass == 0 and g[9]:
                    continue            # SPADE-epilogue variants only serve SPADE ops
                # a SPADE op may also run unfused: plain 1x1 variant (+ split-K) and a modulate kernel
                for ks in ([1] if g[9] else [1, 2, 3, 4, 6, 8, 12, 16]):
                    lib.rib_set_choice(h, B, H, W, name.encode(), vi, ks)
                    t = timeit()
                    if t is None:
                        continue
                    results.append((t, vi, ks))
                    if t < best[0]:
                        best = (t, vi, ks)
            if best[1] is None:
                lib.rib_set_choice(h, B, H, W, name.encode(), -1, 1)
            else:
                lib.rib_set_choice(h, B, H, W, name.encode(), best[1], best[2])
                chosen[name] = geoms[best[1]][:10] + [best[2], geoms[best[1]][10], geoms[best[1]][11]]
            total_before += base; total_after += best[0]
            results.sort()
            report.append({"shape": [B, H, W], "op": name, "default_us": base, "default": tile0, "best_us": best[0],
                           "best": (geoms[best[1]][:10] + [best[2], geoms[best[1]][10], geoms[best[1]][11]]) if best[1] is not None else None,
                           "tflops_best": flops / best[0] / 1e6, "top": [(round(t, 1), geoms[v], k) for t, v, k in results[:4]]})
            print("%-46s default %7.1f us  best %7.1f us (%5.1f TF)  %s" % (
                name[-46:], base, best[0], flops / best[0] / 1e6,
                ("geom %s ksplit %d kw %d tb %d" % (geoms[best[1]][:6], best[2], geoms[best[1]][10], geoms[best[1]][11])) if best[1] is not None else "model choice"), flush=True)
        print("# %dx%d B=%d: %d ops, default %.0f us -> tuned %.0f us (%.1f s)" % (H, W, B, len(ops), total_before, total_after, time.time() - t_start))
        key = "%d,%d,%d" % (B, H, W)
        if args.only is None:
            table[key] = chosen
        else:
            entry = table.setdefault(key, {})
            for name, _, _, _ in ops:
                entry.pop(name, None)          # "model choice" results drop a stale entry
            entry.update(chosen)
    tuning.save(table, args.out)
    if args.report:
        with open(args.report, "w") as f:
            json.dump(report, f, indent=0)


if __name__ == "__main__":
    main()
