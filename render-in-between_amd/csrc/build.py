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
# stage 1 (motion transformer, include/rib_motion.h) is its own small library
MOTION_SRC = os.path.join(HERE, "motion.hip")
MOTION_DEPS = [MOTION_SRC, os.path.join(HERE, "..", "..", "include", "rib_motion.h")]
MOTION_OUT = os.path.join(HERE, "libribmotion.so")


def hipcc_path():
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (need the ROCm toolchain to build librib.so)")


def needs_build(out=OUT, deps=DEPS):
    if not os.path.exists(out):
        return True
    t = os.path.getmtime(out)
    return any(os.path.getmtime(d) > t for d in deps)


def _compile(src, out, verbose):
    cmd = [hipcc_path(), "-O3", "--offload-arch=gfx950", "-std=c++17", "-shared", "-fPIC",
           src, "-o", out + ".tmp"]
    if verbose:
        print("[rib build]", " ".join(cmd), flush=True)
    subprocess.run(cmd, check=True, cwd=HERE)
    os.replace(out + ".tmp", out)


def build(force=False, verbose=True):
    if force or needs_build(MOTION_OUT, MOTION_DEPS):
        _compile(MOTION_SRC, MOTION_OUT, verbose)
    if force or needs_build():
        _compile(SRC, OUT, verbose)
    return OUT


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(OUT)
