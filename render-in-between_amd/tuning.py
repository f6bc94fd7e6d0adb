"""Measured kernel choices (tools/autotune.py) for gfx950: per (B,H,W) and per
launch, the tile geometry {FRW,WM,WN,MF,NF,BK,STRIDE,KS,UPS,SPADE}, split-K (and wave groups KW)
factor that ran fastest on an MI355X.  Shapes without an entry fall back to the
analytic cost model in csrc/rib.hip (choose_variant)."""
from __future__ import annotations

import ctypes as C
import json
import os

TUNING_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "tuning_gfx950.json")
TUNING_PATH_BF16 = os.path.join(os.path.dirname(os.path.abspath(__file__)), "tuning_gfx950_bf16.json")   # measured on the bf16-storage kernels


def load(path=None, bf16=False):
    if path is None:
        path = TUNING_PATH_BF16 if bf16 else TUNING_PATH
    if os.path.exists(path):
        with open(path) as f:
            return json.load(f)
    return {}


def save(table, path=TUNING_PATH):
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    with open(path, "w") as f:
        json.dump(table, f, indent=0, sort_keys=True)


def apply(lib, handle, table, B, H, W, bf16=False):
    """Pin the tuned choices of shape (B,H,W) on a handle; returns how many were applied.  The table
    is measured in fp32; a bf16 handle takes the bf16 twin of each tuned geometry where one exists."""
    entry = table.get("%d,%d,%d" % (B, H, W))
    if not entry:
        return 0
    g11 = (C.c_int * 12)()
    geoms = {}
    for i in range(lib.rib_num_variants()):
        is_bf16 = lib.rib_variant_info(i, g11) == 1
        if is_bf16 == bool(bf16) or (bf16 and tuple(g11) not in geoms):
            geoms[tuple(g11)] = i
    n = 0
    for op, choice in entry.items():
        # entry = geometry[10] + [ksplit] (+ [KW, TB]: wave groups per workgroup, filter slices per
        # barrier; 1 when absent)
        kwg = int(choice[11]) if len(choice) > 11 else 1
        tb = int(choice[12]) if len(choice) > 12 else 1
        idx = geoms.get(tuple(choice[:10]) + (kwg, tb))
        if idx is None and bf16:
            idx = geoms.get(tuple(choice[:10]) + (1, 1))
        if idx is None:
            continue                      # variant table changed since tuning: model choice
        if lib.rib_set_choice(handle, B, H, W, op.encode(), idx, int(choice[10])) == 0:
            n += 1
    return n
