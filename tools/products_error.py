#!/usr/bin/env python3
"""Which fp32 path is closer to the truth?  (VERDICT r05 item 1's reporting rule.)

Runs the generator on the GPU with exact-fp32 products (the default) and with the opt-in split-bf16 products on the plain
GEMMs (rib_set_products(RIB_PRODUCTS_BF16X3)), and compares BOTH with an fp64 evaluation of the CPU oracle on the same
inputs: every tap of the mid_64 case (the 56 intermediates tests/test_gpu_parity.py pins) and the outputs of full_512.
The split mode may only become a default if its max-abs error is <= the exact path's on EVERY row; the JSON says so.

    python tools/products_error.py --out gpurun_out/r06_products_error.json
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch      # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=None)
    args = ap.parse_args()
    import render_in_between_amd as rib
    from render_in_between_amd import synth
    from oracle import generator_ref      # tools/ may use the checker

    mid = dict(num_filters=16, max_num_filters=64, mask=dict(num_filters=32, max_num_filters=64), embed=dict(num_filters=32, max_num_filters=64))
    rows = []

    def err(v, ref):
        d = (v.double().cpu() - ref).abs()
        return float(d.max()), float((d * d).mean().sqrt())

    for case, cfg, seed, in_seed, H, taps_on in (("mid_64", rib.hsm_gen_config(**mid), 7, 7, 64, True), ("full_512", rib.hsm_gen_config(), 0, 123, 512, False)):
        spec = rib.GenSpec.from_cfg(cfg)
        sd = synth.make_state_dict(spec, seed)
        label, fake, prev = synth.make_inputs(spec, 1, H, H, in_seed)
        otaps = {} if taps_on else None
        oimg, omask = generator_ref.RefGenerator(spec, sd, dtype=torch.float64)(label, None, fake, prev, taps=otaps)
        got = {}
        for products in ("f32", "bf16x3"):
            G = rib.Generator(cfg, products=products).eval()
            G.load_state_dict(sd)
            if taps_on:
                G.enable_taps()
            img, mask = G(label, None, fake, prev)
            torch.cuda.synchronize()
            out = {"img": img.cpu(), "mask": mask.cpu()}
            if taps_on:
                out.update({k: v.cpu() for k, v in G.read_taps(1, H, H).items()})
            got[products] = out
            n_x3 = sum(1 for x in G.launch_info(1, H, H) if "gemm (LDS-DMA" in x["tile"])
            del G
        ref = {"img": oimg, "mask": omask}
        if taps_on:
            ref.update(otaps)
        for k in got["f32"]:
            e0, r0 = err(got["f32"][k], ref[k])
            e1, r1 = err(got["bf16x3"][k], ref[k])
            rows.append({"case": case, "tensor": k, "scale": float(ref[k].abs().max()), "f32_max_abs": e0, "f32_rms": r0, "bf16x3_max_abs": e1, "bf16x3_rms": r1,
                         "identical": bool(torch.equal(got["f32"][k], got["bf16x3"][k])), "x3_leq_f32": e1 <= e0})
        ri = [r for r in rows if r["case"] == case and r["tensor"] == "img"][0]
        print("%s: %d launches take the split products; img max-abs vs fp64: f32 %.3e, bf16x3 %.3e" % (case, n_x3, ri["f32_max_abs"], ri["bf16x3_max_abs"]), file=sys.stderr)
    differing = [r for r in rows if not r["identical"]]
    summary = {"rows": len(rows), "rows_where_the_two_paths_differ": len(differing),
               "x3_max_abs_leq_f32_on_every_row": all(r["x3_leq_f32"] for r in rows),
               "rows_where_x3_is_worse": [r["case"] + ":" + r["tensor"] for r in rows if not r["x3_leq_f32"]],
               "median_ratio_rms_x3_over_f32_on_differing_rows": (sorted(r["bf16x3_rms"] / r["f32_rms"] for r in differing)[len(differing) // 2] if differing else None),
               "rule": "the split mode may be a default only if x3_max_abs_leq_f32_on_every_row (VERDICT r05 item 1); otherwise it stays opt-in"}
    doc = {"summary": summary, "rows": rows}
    text = json.dumps(doc, indent=1)
    if args.out:
        with open(args.out, "w") as f:
            f.write(text)
    print(json.dumps(summary))


if __name__ == "__main__":
    main()
