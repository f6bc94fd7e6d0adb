"""Measured kernel choices (tools/autotune.py) for gfx950: per (B,H,W) and per
launch, the tile geometry {FRW,WM,WN,MF,NF,BK,STRIDE,KS,UPS,SPADE}, split-K (and wave groups KW)
factor that ran fastest on an MI355X.  Shapes without an entry fall back to the
analytic cost model in csrc/rib.hip (choose_variant)."""
from __future__ import annotations

import ctypes as C
import json
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
TUNING_PATH = os.path.join(_HERE, "tuning_gfx950.json")
# one table per precision mode, each measured on that mode's own kernels (tools/autotune.py --dtype ...)
# (the half mode runs the bf16 mode's kernels with another element type: same geometries, same table)
TUNING_PATHS = {"f32": TUNING_PATH, "bf16": os.path.join(_HERE, "tuning_gfx950_bf16.json"), "f16": os.path.join(_HERE, "tuning_gfx950_bf16.json")}
PREC = {"f32": 0, "bf16": 1, "f16": 2}           # what rib_variant_info returns for a variant of that mode


def load(path=None, dtype="f32"):
    if path is None:
        # RIB_TUNING_TABLE_<DTYPE>=<file>: another table for this process (A/B of two tables in one gpurun call)
        path = os.environ.get("RIB_TUNING_TABLE_" + dtype.upper()) or TUNING_PATHS[dtype]
    if os.path.exists(path):
        with open(path) as f:
            return json.load(f)
    return {}


def save(table, path=TUNING_PATH):
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    with open(path, "w") as f:
        json.dump(table, f, indent=0, sort_keys=True)


def apply(lib, handle, table, B, H, W, dtype="f32"):
    """Pin the tuned choices of shape (B,H,W) on a handle; returns how many were applied.  Only variants of the
    handle's precision mode are eligible; an entry whose geometry no longer exists is skipped (model choice)."""
    entry = table.get("%d,%d,%d" % (B, H, W))
    if not entry:
        return 0
    g11 = (C.c_int * 12)()
    geoms = {}
    for i in range(lib.rib_num_variants()):
        if lib.rib_variant_info(i, g11) == PREC[dtype]:
            geoms[tuple(g11)] = i
    n = 0
    skipped = []
    for op, choice in entry.items():
        # entry = geometry[10] + [ksplit] (+ [KW, TB]: wave groups per workgroup, filter slices per
        # barrier; 1 when absent)
        kwg = int(choice[11]) if len(choice) > 11 else 1
        tb = int(choice[12]) if len(choice) > 12 else 1
        idx = geoms.get(tuple(choice[:10]) + (kwg, tb))
        if idx is None:
            skipped.append(op)            # variant table changed since tuning: model choice
            continue
        if lib.rib_set_choice(handle, B, H, W, op.encode(), idx, int(choice[10])) == 0:
            n += 1
    last_skipped[(dtype, B, H, W)] = skipped
    if skipped:
        import warnings
        warnings.warn("tuning table %s: %d of %d entries of shape %d,%d,%d name a kernel geometry this library does not have "
                      "(%s ...): those launches take the cost model's choice; re-run tools/autotune.py for this build"
                      % (dtype, len(skipped), len(entry), B, H, W, skipped[0]))
    return n


last_skipped = {}           # (dtype, B, H, W) -> op names whose tabled geometry the loaded library lacks (apply())
