#!/usr/bin/env python3
"""HBM traffic per kernel class from two rocprofv3 --pmc passes (FETCH_SIZE and WRITE_SIZE cannot
share a pass on gfx950: MI355X_MICROARCH.md 'rocprofv3 PMC slots').

    rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_f -- python3 tools/prof_ops.py --run
    rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_w -- python3 tools/prof_ops.py --run
    python3 tools/pmc_traffic.py gpurun_out/pmc_f gpurun_out/pmc_w profiles/r01_pmc_traffic.json

Corrections (MI355X_MICROARCH.md §HBM): both counters are in KiB; on gfx950 FETCH_SIZE reports
exactly half of the bytes of a wide (16 B/lane) coalesced streaming read, which is how every
operand of these kernels is loaded, so it is doubled; WRITE_SIZE is exact.
"""
import csv, glob, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools.prof_ops import plan_names

NAMES = ("igemm", "spade", "stats", "pool", "eltwise", "pack", "conv_aux")


def per_op(d, counter, n):
    f = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)[0]
    disp = {}
    for r in csv.DictReader(open(f)):
        if not r["Kernel_Name"].startswith(("void rib::", "rib::")) or r["Counter_Name"] != counter:
            continue
        disp[int(r["Dispatch_Id"])] = disp.get(int(r["Dispatch_Id"]), 0.0) + float(r["Counter_Value"])
    ids = sorted(disp)
    steps = len(ids) // n
    ids = ids[-n * steps:]
    out = [0.0] * n
    for i, k in enumerate(ids):
        out[i % n] += disp[k] / steps
    return out, steps


def main():
    df, dw, outp = sys.argv[1:4]
    ops = plan_names(1, 512, 512)
    n = len(ops)
    fetch, steps = per_op(df, "FETCH_SIZE", n)
    write, _ = per_op(dw, "WRITE_SIZE", n)
    cls = {}
    for o, f, w in zip(ops, fetch, write):
        c = cls.setdefault(NAMES[o["class"]], {"launches": 0, "fetch_bytes_raw": 0.0, "write_bytes": 0.0})
        c["launches"] += 1
        c["fetch_bytes_raw"] += f * 1024.0
        c["write_bytes"] += w * 1024.0
    for c in cls.values():
        c["hbm_bytes_corrected"] = 2.0 * c["fetch_bytes_raw"] + c["write_bytes"]
        c["hbm_bytes_per_launch"] = c["hbm_bytes_corrected"] / c["launches"]
    res = {"workload": "512x512 B=1 fp32 forward + blend", "steps_averaged": steps, "classes": cls,
           "total_hbm_bytes_per_step": sum(c["hbm_bytes_corrected"] for c in cls.values()),
           "correction": "FETCH_SIZE x2 (gfx950 wide-read under-count), KiB -> bytes; WRITE_SIZE exact"}
    with open(outp, "w") as f:
        json.dump(res, f, indent=1)
    print(json.dumps({k: round(v["hbm_bytes_corrected"] / 1e6, 1) for k, v in cls.items()}), "MB/step; total %.1f MB" % (res["total_hbm_bytes_per_step"] / 1e6))


if __name__ == "__main__":
    main()
