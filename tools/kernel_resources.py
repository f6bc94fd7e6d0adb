"""Parse `hipcc -Rpass-analysis=kernel-resource-usage` remarks of the librib.so build into a table
(VGPRs, AGPRs, scratch, occupancy, LDS per k_igemm instantiation).

    hipcc -O3 --offload-arch=gfx950 -std=c++17 -shared -fPIC rib.hip -o /tmp/x.so \
          -Rpass-analysis=kernel-resource-usage 2> /tmp/res.txt
    python tools/kernel_resources.py /tmp/res.txt [filter]
"""
import re
import subprocess
import sys


def main():
    t = open(sys.argv[1]).read()
    flt = sys.argv[2] if len(sys.argv) > 2 else ""
    for b in t.split("Function Name: ")[1:]:
        name = b.split("\n")[0].split(" [-Rpass")[0].strip()

        def g(k):
            m = re.search(k + r": (\d+)", b)
            return int(m.group(1)) if m else -1
        d = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
        d = d.replace("rib::", "").replace("(anonymous namespace)::", "")
        if flt and flt not in d:
            continue
        print("%-95s vgpr %3d agpr %3d scratch %4d occ %2d lds %6d" % (
            d[:95], g("VGPRs"), g("AGPRs"), g(r"ScratchSize \[bytes/lane\]"), g(r"Occupancy \[waves/SIMD\]"),
            g(r"LDS Size \[bytes/block\]")))


if __name__ == "__main__":
    main()
