#!/usr/bin/env python3
"""Compile every k_igemm shard of two source trees with -Rpass-analysis=kernel-resource-usage and list the
instantiations whose occupancy (waves/SIMD, from VGPRs + AGPRs and LDS) differs.  A few registers or a few KB of LDS
across a boundary (128 registers, 160 KB / n of LDS) cost 5-20 % on the launches that use the variant; this round lost
8 % of fp32 frame time that way before it was noticed.

    python3 tools/occupancy_diff.py <csrc dir A> <csrc dir B>
"""
import os, re, subprocess, sys, tempfile
from concurrent.futures import ThreadPoolExecutor


def resources(csrc):
    out = tempfile.mkdtemp()
    def one(s):
        r = subprocess.run(["hipcc", "-O3", "--offload-arch=gfx950", "-std=c++17", "-fPIC", "-DRIB_SECTION=%d" % s, "-c", "igemm_shard.hip",
                            "-o", os.path.join(out, "s%d.o" % s), "-Rpass-analysis=kernel-resource-usage"], cwd=csrc, capture_output=True, text=True)
        return r.stderr
    with ThreadPoolExecutor(8) as ex:
        text = "".join(ex.map(one, range(8)))
    res = {}
    for b in text.split("Function Name: ")[1:]:
        name = b.split("\n")[0].split(" [-Rpass")[0].strip()
        g = lambda k: int(re.search(k + r": (\d+)", b).group(1))
        res[name] = (g("VGPRs"), g("AGPRs"), g(r"Occupancy \[waves/SIMD\]"), g(r"LDS Size \[bytes/block\]"), g(r"ScratchSize \[bytes/lane\]"))
    names = list(res)
    dm = subprocess.run(["c++filt"] + names, capture_output=True, text=True).stdout.strip().split("\n")
    norm = {}
    for n, d in zip(names, dm):
        m = re.match(r"void rib::k_igemm<(.*)>\(", d)
        if not m:
            continue
        a = [x.strip() for x in m.group(1).split(",")]
        a = a + ["0", "true", "true", "1", "1"][len(a) - 10:] if len(a) < 15 else a
        a[10] = {"false": "0", "true": "1"}.get(a[10], a[10])
        norm["k_igemm<" + ", ".join(a) + ">"] = res[n]
    return norm


def main():
    a, b = resources(sys.argv[1]), resources(sys.argv[2])
    common = sorted(set(a) & set(b))
    worse = [k for k in common if b[k][2] < a[k][2]]
    better = [k for k in common if b[k][2] > a[k][2]]
    print("%d / %d instantiations, %d common; occupancy worse in B: %d, better: %d; scratch users in B: %d" % (
        len(a), len(b), len(common), len(worse), len(better), sum(1 for v in b.values() if v[4] > 0)))
    for k in worse:
        print("WORSE  %-78s (vgpr, agpr, occ, lds) %s -> %s" % (k, a[k][:4], b[k][:4]))
    for k in better:
        print("BETTER %-78s %s -> %s" % (k, a[k][:4], b[k][:4]))


if __name__ == "__main__":
    main()
