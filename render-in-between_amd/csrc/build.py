"""Build librib.so (HIP kernels + C-ABI runtime) for gfx950 in-tree with hipcc.

    python render-in-between_amd/csrc/build.py [--force]

hipcc cross-compiles without a GPU; the built .so is git-ignored but travels to
the GPU box with the gpurun snapshot.
"""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "rib.hip")
DEPS = [SRC, os.path.join(HERE, "kernels.hip.h"), os.path.join(HERE, "raster.hip.h"),
        os.path.join(HERE, "..", "..", "include", "rib.h")]
OUT = os.path.join(HERE, "librib.so")


def hipcc_path():
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (need the ROCm toolchain to build librib.so)")


def needs_build():
    if not os.path.exists(OUT):
        return True
    t = os.path.getmtime(OUT)
    return any(os.path.getmtime(d) > t for d in DEPS)


def build(force=False, verbose=True):
    if not force and not needs_build():
        return OUT
    cmd = [hipcc_path(), "-O3", "--offload-arch=gfx950", "-std=c++17", "-shared", "-fPIC",
           SRC, "-o", OUT + ".tmp"]
    if verbose:
        print("[rib build]", " ".join(cmd), flush=True)
    subprocess.run(cmd, check=True, cwd=HERE)
    os.replace(OUT + ".tmp", OUT)
    return OUT


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(OUT)
