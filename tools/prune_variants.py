#!/usr/bin/env python3
"""Rewrite csrc/variants.def without the k_igemm entries no launch plan uses (tools/variant_usage.py) and deal the
remaining entries to the shard objects (RIB_NSECTIONS, variants.hip.h) by estimated compile cost.  Tuning tables name geometries, not indices, so
they stay valid; an entry that a later tuning run might have liked is simply no longer offered.

    python tools/prune_variants.py [--dry-run]
"""
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tools.variant_usage import usage          # noqa: E402

DEF = os.path.join(ROOT, "render-in-between_amd", "csrc", "variants.def")
ROWS = {"RIB_VB": 2, "RIB_VBX": 2}                                    # variant rows (rib_variant_info) per entry; default 1
INST = {"RIB_V": 3, "RIB_VK": 3, "RIB_VT": 3, "RIB_VTK": 3, "RIB_V9": 3, "RIB_VU4": 2, "RIB_VS": 1, "RIB_VSK": 1, "RIB_VB": 2,
        "RIB_VBX": 2, "RIB_V1D": 2, "RIB_VS1D": 1, "RIB_VD": 2, "RIB_VSD": 1, "RIB_VD9": 1}      # kernel instantiations per entry


def main():
    info, used = usage()
    lines = open(DEF).read().split("\n")
    out, idx, dropped, kept = [], 0, 0, []
    for l in lines:
        m = re.match(r"(RIB_V[A-Z0-9]*)\((\d+),", l)
        if not m:
            out.append(l)
            continue
        n = ROWS.get(m.group(1), 1)
        if any((idx + k) in used for k in range(n)):
            kept.append(len(out))
            out.append(l)
        else:
            dropped += 1
        idx += n
    # sections: heaviest entries first onto the lightest shard
    with open(os.path.join(os.path.dirname(DEF), "variants.hip.h")) as f:
        nsec = int(re.search(r"#define RIB_NSECTIONS (\d+)", f.read()).group(1))
    load = [0] * nsec

    def weight(line):      # compile cost: instantiations; the fully unrolled phase convolutions (UPS) take ~4x as long each
        macro = re.match(r"(RIB_V[A-Z0-9]*)\(", line).group(1)
        return INST[macro] * (4 if (macro == "RIB_VU4" or ", true," in line) else 1)
    order = sorted(kept, key=lambda i: -weight(out[i]))
    for i in order:
        s = load.index(min(load))
        load[s] += weight(out[i])
        out[i] = re.sub(r"^(RIB_V[A-Z0-9]*)\(\d+,", r"\1(%d," % s, out[i])
    # comment blocks that lost all their entries
    text = "\n".join(out)
    print("entries kept %d, dropped %d; instantiations per shard %s" % (len(kept), dropped, load))
    if "--dry-run" not in sys.argv:
        open(DEF, "w").write(text)


if __name__ == "__main__":
    main()
