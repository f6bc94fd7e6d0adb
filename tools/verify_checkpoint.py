#!/usr/bin/env python3
"""Day-one check of a real generator checkpoint in ONE command.

    python tools/verify_checkpoint.py /path/to/netG_epoch006.pth [--out tests/golden/real_ckpt] [--sizes 64 128 256]

The published checkpoint (PGNR/configs/HSM.yaml:2, README.md:42-48) is not in the tree and there is no network here; every
code path it touches has only ever seen seed-defined checkpoints of the reference's exact key set.  This tool is what to
run the moment the file is available.  It exits non-zero on the first hard failure and prints a JSON report; steps that
need something the machine lacks are reported as "skipped", never silently passed.

  1 load      torch.load + the ['state_dict'] / 'module.' handling of PGNR/utils/utils.py:107-119; strict key-set and shape
              comparison with the reference's own 372 keys (tests/golden/state_dict_keys.json)                   [CPU]
  2 fold      the library's eval-mode spectral-norm fold (C++, host-only handle) against the oracle's, every convolution;
              range of the FOLDED filters: the largest |W / sigma| per layer, and whether IEEE half (65504) holds them  [CPU]
  3 goldens   where /root/reference is present (the build container): the imported REFERENCE generator on this checkpoint
              with seed-defined inputs; its outputs are stored under --out (numbers only) with the oracle-vs-reference
              distance, exactly as tests/golden/make_golden.py does for the synthetic checkpoints           [CPU, build box]
  4 gpu       where a GPU is present: the HIP path in f32 (<= 2e-4 of the oracle; the north star's bar is 1e-3), bf16 and f16
              (finite outputs, error against the oracle, against the precision model of oracle/precision_model.py);
              an f16 overflow is reported as a FAILURE of that mode only (the checkpoint then runs in bf16 / f32)     [GPU]
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np      # noqa: E402
import torch            # noqa: E402


def load_checkpoint(path, trust_pickle=False):
    """PGNR/utils/utils.py:107-119: missing file -> ValueError; optional 'state_dict' wrapper; 'module.' prefixes stripped.
    torch.load runs with weights_only=True (tensors and plain containers only) unless --trust-pickle is given: the reference's
    own torch.load (torch 1.4) unpickles arbitrary objects, which is only acceptable for a file whose origin is trusted."""
    if not os.path.exists(path):
        raise ValueError("No checkpoint found at {}".format(path))
    try:
        sd = torch.load(path, map_location="cpu", weights_only=not trust_pickle)
    except Exception as e:      # noqa: BLE001
        if trust_pickle:
            raise
        raise RuntimeError("torch.load(weights_only=True) refused the file (%s: %s); if its origin is trusted, re-run with --trust-pickle"
                           % (type(e).__name__, " ".join(str(e).split())[:200]))
    if "state_dict" in sd and not torch.is_tensor(sd["state_dict"]):
        sd = sd["state_dict"]
    return {k.replace("module.", ""): v.detach().to(torch.float32) for k, v in sd.items()}


def step_load(sd, report):
    with open(os.path.join(ROOT, "tests", "golden", "state_dict_keys.json")) as f:
        want = json.load(f)
    want = {k: tuple(v) for k, v in (want.items() if isinstance(want, dict) else want)}
    missing = sorted(set(want) - set(sd))
    unexpected = sorted(set(sd) - set(want))
    shapes = [(k, tuple(sd[k].shape), want[k]) for k in want if k in sd and tuple(sd[k].shape) != want[k]]
    nonfinite = sorted(k for k, v in sd.items() if not bool(torch.isfinite(v).all()))
    report["load"] = {"tensors": len(sd), "expected": len(want), "missing": missing[:8], "unexpected": unexpected[:8],
                      "shape_mismatches": [list(map(str, s)) for s in shapes[:8]], "non_finite_tensors": nonfinite[:8],
                      "ok": not (missing or unexpected or shapes or nonfinite)}
    return report["load"]["ok"]


def step_fold(sd, report):
    import ctypes as C
    import render_in_between_amd as rib
    from render_in_between_amd import _native
    from oracle import generator_ref      # tools/ may use the checker
    lib = _native.lib()
    spec = rib.GenSpec.from_cfg(rib.hsm_gen_config())
    c = _native.RibConfig(**{n: getattr(spec, n) for n, _ in _native.RibConfig._fields_})
    h = C.c_void_p()
    assert lib.rib_create(C.byref(c), -1, C.byref(h)) == 0       # host-only handle: inventory, fold, plans
    try:
        for k, v in sd.items():
            t = v.contiguous()
            dims = (C.c_int64 * max(t.dim(), 1))(*t.shape)
            rc = lib.rib_set_tensor(h, k.encode(), C.c_void_p(t.data_ptr()), t.dim(), dims)
            if rc != 0:
                report["fold"] = {"ok": False, "error": "rib_set_tensor(%s): %s" % (k, lib.rib_last_error(h).decode())}
                return False
        rc = lib.rib_finalize_weights(h)
        if rc != 0:
            report["fold"] = {"ok": False, "error": "rib_finalize_weights: " + lib.rib_last_error(h).decode()}
            return False
        worst, rng = 0.0, []
        convs = sorted({k[:-len(".layers.conv.bias")] for k in sd if k.endswith(".layers.conv.bias") and ".norm.mlps." not in k})
        for name in convs:
            if name.startswith("label_embedding.") or name == "conv_mask":
                continue                                         # never called by forward: not folded
            w, b = generator_ref.conv_weight(sd, name)
            got_w = torch.empty_like(w)
            got_b = torch.empty_like(b)
            if lib.rib_debug_conv_weight(h, name.encode(), C.c_void_p(got_w.data_ptr()), C.c_void_p(got_b.data_ptr())) != 0:
                report["fold"] = {"ok": False, "error": "rib_debug_conv_weight(%s): %s" % (name, lib.rib_last_error(h).decode())}
                return False
            scale = max(1.0, float(w.abs().max()))
            worst = max(worst, float((got_w - w).abs().max()) / scale, float((got_b - b).abs().max()))
            rng.append((float(w.abs().max()), name))
        spade = [(float(v.abs().max()), k[:-len(".weight")]) for k, v in sd.items() if ".norm.mlps." in k and k.endswith(".weight")]
        rng.sort(reverse=True)
        spade.sort(reverse=True)
        top = max(rng[0][0], spade[0][0] if spade else 0.0)
        report["fold"] = {"convolutions": len(rng), "max_rel_diff_vs_oracle_fold": worst,
                          "largest_folded_filter_values": [{"layer": n, "max_abs": v} for v, n in rng[:5]],
                          "largest_gamma_beta_filter_values": [{"layer": n, "max_abs": v} for v, n in spade[:3]],
                          "fits_ieee_half": top <= 65504.0, "ok": worst <= 1e-6}
        return report["fold"]["ok"]
    finally:
        lib.rib_destroy(h)


def inputs_for(spec, size, seed=0):
    from render_in_between_amd import synth
    return synth.make_inputs(spec, 1, size, size, seed)


def step_goldens(sd, sizes, out_dir, report):
    if not os.path.isdir("/root/reference"):
        report["goldens"] = {"skipped": "no /root/reference on this machine (run in the build container to store the reference's outputs)"}
        return True
    import render_in_between_amd as rib
    from oracle import generator_ref, ref_import
    cfg = rib.hsm_gen_config()
    spec = rib.GenSpec.from_cfg(cfg)
    G = ref_import.load_reference_generator(cfg)
    G.load_state_dict(sd, strict=True)               # the reference's own strict load of the same file
    os.makedirs(out_dir, exist_ok=True)
    rows, ok = {}, True
    for size in sizes:
        label, fake, prev = inputs_for(spec, size)
        with torch.no_grad():
            rimg, rmask = G(label, None, fake, prev)
        oimg, omask = generator_ref.RefGenerator(spec, sd)(label, None, fake, prev)
        d = (float((rimg - oimg).abs().max()), float((rmask - omask).abs().max()))
        sat = float((rimg.abs() > 0.999).float().mean())
        np.savez_compressed(os.path.join(out_dir, "real_%d.npz" % size), img=rimg.numpy(), mask=rmask.numpy())
        rows[str(size)] = {"oracle_vs_reference": {"img": d[0], "mask": d[1]}, "tanh_saturated_fraction": sat,
                           "img_mean": float(rimg.double().mean()), "mask_mean": float(rmask.double().mean())}
        ok = ok and d[0] <= 1e-4 and d[1] <= 1e-4
    with open(os.path.join(out_dir, "real_report.json"), "w") as f:
        json.dump(rows, f, indent=1)
    report["goldens"] = {"stored_under": out_dir, "cases": rows, "ok": ok}
    return ok


def step_gpu(sd, sizes, out_dir, report):
    if not torch.cuda.is_available():
        report["gpu"] = {"skipped": "no GPU on this machine"}
        return True
    import render_in_between_amd as rib
    from oracle import generator_ref, precision_model
    cfg = rib.hsm_gen_config()
    spec = rib.GenSpec.from_cfg(cfg)
    R = generator_ref.RefGenerator(spec, sd)
    rows, ok = {}, True
    for dtype in ("f32", "bf16", "f16"):
        try:
            G = rib.Generator(cfg, compute_dtype=dtype).eval()
            G.load_state_dict(sd)                     # strict; f16: range check of the folded filters + finite probe
        except Exception as e:      # noqa: BLE001 (a mode that cannot take the checkpoint is a finding, not a crash)
            rows[dtype] = {"ok": dtype != "f32", "refused": "%s: %s" % (type(e).__name__, e)}
            ok = ok and dtype != "f32"
            continue
        per = {}
        for size in sizes:
            label, fake, prev = inputs_for(spec, size)
            img, mask = [t.cpu() for t in G(label, None, fake, prev)]
            oimg, omask = R(label, None, fake, prev)
            r = {"finite": bool(torch.isfinite(img).all() and torch.isfinite(mask).all()),
                 "max_abs_img": float((img - oimg).abs().max()), "max_abs_mask": float((mask - omask).abs().max()),
                 "mean_abs_img": float((img - oimg).abs().mean()), "mean_abs_mask": float((mask - omask).abs().mean())}
            gold = os.path.join(out_dir, "real_%d.npz" % size)
            if os.path.exists(gold):                  # the reference's own outputs, when step 3 has stored them
                g = np.load(gold)
                r["max_abs_img_vs_reference"] = float(np.abs(img.numpy() - g["img"]).max())
                r["max_abs_mask_vs_reference"] = float(np.abs(mask.numpy() - g["mask"]).max())
            if dtype == "f32":
                r["ok"] = r["finite"] and r["max_abs_img"] <= 2e-4 and r["max_abs_mask"] <= 2e-4
            else:
                m = precision_model.predict(spec, sd, label, fake, prev, fmt=dtype)
                r["model_mean_abs_img"] = m["mean_abs_img"]
                r["mean_abs_img_over_model"] = r["mean_abs_img"] / max(m["mean_abs_img"], 1e-12)
                r["ok"] = r["finite"] and 0.7 <= r["mean_abs_img_over_model"] <= 1.3
            per[str(size)] = r
        rows[dtype] = {"ok": all(v["ok"] for v in per.values()), "sizes": per}
        ok = ok and (rows[dtype]["ok"] or dtype == "f16")      # an f16 failure only rules that mode out for this checkpoint
        del G
    report["gpu"] = {"modes": rows, "ok": ok}
    return ok


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("checkpoint")
    ap.add_argument("--out", default=os.path.join(ROOT, "tests", "golden", "real_ckpt"))
    ap.add_argument("--sizes", type=int, nargs="+", default=[64, 128, 256])
    ap.add_argument("--report", default=None)
    ap.add_argument("--trust-pickle", action="store_true", help="torch.load with weights_only=False (a checkpoint that holds more than tensors and plain containers)")
    args = ap.parse_args(argv)
    report = {"checkpoint": args.checkpoint}
    ok = True
    try:
        sd = load_checkpoint(args.checkpoint, args.trust_pickle)
    except Exception as e:      # noqa: BLE001
        report["load"] = {"ok": False, "error": "%s: %s" % (type(e).__name__, e)}
        ok = False
    else:
        for step in (step_load, step_fold):
            ok = step(sd, report)
            if not ok:
                break
        if ok:
            ok = step_goldens(sd, args.sizes, args.out, report) and ok
            ok = step_gpu(sd, args.sizes, args.out, report) and ok
    report["ok"] = bool(ok)
    text = json.dumps(report, indent=1)
    if args.report:
        with open(args.report, "w") as f:
            f.write(text)
    print(text)
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
