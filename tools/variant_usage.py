#!/usr/bin/env python3
"""Which k_igemm instantiations (csrc/variants.def) does any launch plan use?  CPU only (host-only handles).

    python tools/variant_usage.py            # table: variant index, geometry, how many (config, dtype, shape, op) use it
    python tools/variant_usage.py --unused   # the geometries nothing uses: candidates for removal from variants.def

Plans are built for the two generator configurations of the tests (HSM.yaml and the narrower "mid" one), the
precision modes, every shape of the measured tables plus a sweep of other sizes and batches - with the measured choices
pinned where a table exists (what Generator does) and with the analytic cost model alone (what an untabled shape gets).
"""
import argparse
import ctypes as C
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import render_in_between_amd as rib                        # noqa: E402
from render_in_between_amd import _native, tuning         # noqa: E402

CONFIGS = {
    "full": {},
    "mid": dict(num_filters=16, max_num_filters=64, mask=dict(num_filters=32, max_num_filters=64), embed=dict(num_filters=32, max_num_filters=64)),
}
SWEEP = [(b, h, w) for b in (1, 2, 3, 4, 8) for h, w in ((32, 32), (32, 48), (64, 64), (64, 96), (128, 128), (256, 256), (320, 480), (512, 512), (720, 1280), (1024, 1024))]


def usage():
    lib = _native.lib()
    g = (C.c_int * 12)()
    info = [(lib.rib_variant_info(i, g), tuple(g)) for i in range(lib.rib_num_variants())]
    used = {}
    buf = C.create_string_buffer(512)
    for cname, over in CONFIGS.items():
        spec = rib.GenSpec.from_cfg(rib.hsm_gen_config(**over))
        c = _native.RibConfig(**{n: getattr(spec, n) for n, _ in _native.RibConfig._fields_})
        for dtype in ("f32", "bf16", "f16"):
            table = tuning.load(dtype=dtype)
            shapes = sorted(set(SWEEP) | {tuple(int(v) for v in k.split(",")) for k in table})
            for tuned in (True, False):
                h = C.c_void_p()
                assert lib.rib_create(C.byref(c), -1, C.byref(h)) == 0
                assert lib.rib_set_compute_dtype(h, {"f32": 0, "bf16": 1, "f16": 3}[dtype]) == 0
                for (B, H, W) in shapes:
                    if tuned and cname == "full":
                        tuning.apply(lib, h, table, B, H, W, dtype=dtype)
                    n = lib.rib_num_launches(h, B, H, W)
                    assert n > 0, (cname, dtype, B, H, W, lib.rib_last_error(h))
                    for i in range(n):
                        lib.rib_debug_launch_info(h, B, H, W, i, buf, 512)
                        m = re.search(r" v(\d+)\|", buf.value.decode())
                        if m:
                            used.setdefault(int(m.group(1)), set()).add((cname, dtype, B, H, W, tuned))
                lib.rib_destroy(h)
    return info, used


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--unused", action="store_true")
    a = ap.parse_args()
    info, used = usage()
    names = {0: "f32", 1: "bf16", 2: "f16"}
    for i, (prec, g) in enumerate(info):
        n = len(used.get(i, ()))
        if a.unused and n:
            continue
        if not a.unused and not n:
            continue
        print("v%-3d %-4s FRW %2d WM %d WN %d MF %d NF %d BK %2d S %d KS %d UPS %d SPADE %d KW %d TB %3d   uses %d" % ((i, names[prec]) + g + (n,)))
    print("# %d variants, %d used by some plan, %d by none" % (len(info), len(used), len(info) - len(used)))


if __name__ == "__main__":
    main()
