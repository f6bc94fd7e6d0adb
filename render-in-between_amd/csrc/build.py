"""Build librib.so (HIP kernels + C-ABI runtime) and libribmotion.so for gfx950 in-tree with hipcc.

    python render-in-between_amd/csrc/build.py [--force] [--jobs N]

hipcc cross-compiles without a GPU; the built .so files are git-ignored but travel to the GPU box with
the gpurun snapshot.  librib.so is linked from rib.o (runtime, C ABI, the small kernels) and eight
igemm_shard_<s>.o objects, each holding one section of the k_igemm tile variants (variants.def): the
~330 kernel instantiations dominate the build and compile as parallel jobs (about 1 min on 8 cores
instead of 4 min as one translation unit).  Objects are rebuilt only when one of their inputs changed.
"""
import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
INC = os.path.join(HERE, "..", "..", "include")
OBJ = os.path.join(HERE, "build")
NSECTIONS = 8
SRC = os.path.join(HERE, "rib.hip")
SHARD_SRC = os.path.join(HERE, "igemm_shard.hip")
SHARD_DEPS = [SHARD_SRC] + [os.path.join(HERE, f) for f in ("kernels.hip.h", "variants.hip.h", "variants.def")]
DEPS = [SRC, os.path.join(HERE, "raster.hip.h"), os.path.join(INC, "rib.h")] + SHARD_DEPS[1:]
OUT = os.path.join(HERE, "librib.so")
# stage 1 (motion transformer, include/rib_motion.h) is its own small library
MOTION_SRC = os.path.join(HERE, "motion.hip")
MOTION_DEPS = [MOTION_SRC, os.path.join(INC, "rib_motion.h")]
MOTION_OUT = os.path.join(HERE, "libribmotion.so")
FLAGS = ["-O3", "--offload-arch=gfx950", "-std=c++17", "-fPIC"]


def hipcc_path():
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (need the ROCm toolchain to build librib.so)")


def needs_build(out=OUT, deps=DEPS + SHARD_DEPS):
    if not os.path.exists(out):
        return True
    t = os.path.getmtime(out)
    return any(os.path.getmtime(d) > t for d in deps)


def _run(cmd, verbose):
    if verbose:
        print("[rib build]", " ".join(os.path.relpath(c, HERE) if os.path.isabs(c) and c.startswith(HERE) else c for c in cmd), flush=True)
    subprocess.run(cmd, check=True, cwd=HERE)


def _compile_shared(src, out, verbose):
    _run([hipcc_path()] + FLAGS + ["-shared", src, "-o", out + ".tmp"], verbose)
    os.replace(out + ".tmp", out)


def _compile_obj(src, obj, defs, verbose):
    _run([hipcc_path()] + FLAGS + defs + ["-c", src, "-o", obj + ".tmp"], verbose)
    os.replace(obj + ".tmp", obj)
    return obj


def build(force=False, verbose=True, jobs=None):
    if force or needs_build(MOTION_OUT, MOTION_DEPS):
        _compile_shared(MOTION_SRC, MOTION_OUT, verbose)
    if force or needs_build():
        os.makedirs(OBJ, exist_ok=True)
        work = []
        rib_o = os.path.join(OBJ, "rib.o")
        if force or needs_build(rib_o, DEPS):
            work.append((SRC, rib_o, []))
        objs = [rib_o]
        for s in range(NSECTIONS):
            o = os.path.join(OBJ, "igemm_shard_%d.o" % s)
            objs.append(o)
            if force or needs_build(o, SHARD_DEPS):
                work.append((SHARD_SRC, o, ["-DRIB_SECTION=%d" % s]))
        jobs = jobs or max(1, min(len(work), os.cpu_count() or 1))
        with ThreadPoolExecutor(jobs) as ex:
            for f in [ex.submit(_compile_obj, s, o, d, verbose) for s, o, d in work]:
                f.result()
        _run([hipcc_path(), "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-o", OUT + ".tmp"], verbose)
        os.replace(OUT + ".tmp", OUT)
    return OUT


if __name__ == "__main__":
    j = int(sys.argv[sys.argv.index("--jobs") + 1]) if "--jobs" in sys.argv else None
    build(force="--force" in sys.argv, jobs=j)
    print(OUT)
