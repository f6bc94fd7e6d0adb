"""Build librib.so (HIP kernels + C-ABI runtime) and libribmotion.so for gfx950 in-tree with hipcc.

    python render-in-between_amd/csrc/build.py [--force] [--jobs N] [--check]

hipcc cross-compiles without a GPU; the built .so files are git-ignored but travel to the GPU box with
the gpurun snapshot.  librib.so is linked from rib.o (runtime, C ABI, the small kernels) and RIB_NSECTIONS (variants.hip.h: 24)
igemm_shard_<s>.o objects, each holding one section of the k_igemm tile variants (variants.def): the
kernel instantiations dominate the build and compile as a queue of parallel jobs (about 2 min on 8 cores).

Build stamps.  Every object is compiled with -DRIB_BUILD_STAMP="<hash>" where <hash> is the sha256 over the CONTENT
of the sources that object is made from plus the compiler flags and the compiler's version line, and keeps it as a
string ("rib-stamp <tag> <hash>").  An object is rebuilt when the stamp it carries differs from the stamp of the tree
(content, not mtime: a `git checkout` or a touched file cannot leave a stale object behind), the library is re-linked
when it does not carry exactly the objects' stamps, and `rib_build_info()` (include/rib.h) reports them at run time:
bench.py prints the string in its JSON line and tests/test_native_host.py asserts stamp(librib.so) == stamp(tree) for
all objects, so a measured binary that the tracked source does not build cannot go unnoticed.
`--check` prints the tree's and the library's stamps and exits 1 on a mismatch without building.
"""
import hashlib
import os
import re
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
INC = os.path.join(HERE, "..", "..", "include")
OBJ = os.path.join(HERE, "build")


def _nsections():
    with open(os.path.join(HERE, "variants.hip.h")) as f:
        return int(re.search(r"#define RIB_NSECTIONS (\d+)", f.read()).group(1))


NSECTIONS = _nsections()        # shard objects (variants.hip.h); more than cores: the compiles run as a job queue
SRC = os.path.join(HERE, "rib.hip")
SHARD_SRC = os.path.join(HERE, "igemm_shard.hip")
SHARD_DEPS = [SHARD_SRC] + [os.path.join(HERE, f) for f in ("igemm.hip.h", "variants.hip.h", "variants.def")]
DEPS = [SRC, os.path.join(HERE, "kernels.hip.h"), os.path.join(HERE, "raster.hip.h"), os.path.join(INC, "rib.h")] + SHARD_DEPS[1:]
OUT = os.path.join(HERE, "librib.so")
# stage 1 (motion transformer, include/rib_motion.h) is its own small library
MOTION_SRC = os.path.join(HERE, "motion.hip")
MOTION_DEPS = [MOTION_SRC, os.path.join(INC, "rib_motion.h")]
MOTION_OUT = os.path.join(HERE, "libribmotion.so")
FLAGS = ["-O3", "--offload-arch=gfx950", "-std=c++17", "-fPIC"]
LINK_FLAGS = ["--offload-arch=gfx950", "-shared", "-fPIC"]
STAMP_RE = re.compile(rb"rib-stamp ([a-z0-9]+) ([0-9a-f]{16})")


def hipcc_path():
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (need the ROCm toolchain to build librib.so)")


_compiler_id = None


def compiler_id():
    """The line of `hipcc --version` that names the HIP / clang build (part of every stamp)."""
    global _compiler_id
    if _compiler_id is None:
        out = subprocess.run([hipcc_path(), "--version"], capture_output=True, text=True).stdout
        lines = [l.strip() for l in out.splitlines() if "HIP version" in l or "clang version" in l]
        _compiler_id = " | ".join(lines) or "unknown"
    return _compiler_id


def stamp_of(deps, extra=()):
    """sha256 over (file name, content) of deps, the flags and the compiler id: 16 hex digits."""
    h = hashlib.sha256()
    for d in deps:
        h.update(os.path.basename(d).encode() + b"\0")
        with open(d, "rb") as f:
            h.update(f.read())
        h.update(b"\0")
    h.update(" ".join(list(FLAGS) + list(extra)).encode() + b"\0" + compiler_id().encode())
    return h.hexdigest()[:16]


def tree_stamps():
    """{tag: stamp} the tracked sources build to: 'lib' (rib.o: everything), 'shard0'..'shard7', 'motion'."""
    st = {"lib": stamp_of(DEPS + [SHARD_SRC]), "motion": stamp_of(MOTION_DEPS)}
    shard = stamp_of(SHARD_DEPS)
    for s in range(NSECTIONS):
        st["shard%d" % s] = shard
    return st


def embedded_stamps(path):
    """{tag: stamp} of the "rib-stamp <tag> <hash>" strings a built object / library carries."""
    if not os.path.exists(path):
        return {}
    with open(path, "rb") as f:
        data = f.read()
    return {m.group(1).decode(): m.group(2).decode() for m in STAMP_RE.finditer(data)}


def needs_build(out=OUT):
    """Does `out` (librib.so / libribmotion.so) differ from what the tree builds to?"""
    want = tree_stamps()
    have = embedded_stamps(out)
    tags = ["motion"] if os.path.basename(out) == os.path.basename(MOTION_OUT) else [t for t in want if t != "motion"]
    return any(have.get(t) != want[t] for t in tags)


def _run(cmd, verbose):
    if verbose:
        print("[rib build]", " ".join(os.path.relpath(c, HERE) if os.path.isabs(c) and c.startswith(HERE) else c for c in cmd), flush=True)
    subprocess.run(cmd, check=True, cwd=HERE)


def _stamp_def(stamp):
    return ['-DRIB_BUILD_STAMP="%s"' % stamp]


def _compile_shared(src, out, defs, verbose):
    _run([hipcc_path()] + FLAGS + defs + ["-shared", src, "-o", out + ".tmp"], verbose)
    os.replace(out + ".tmp", out)


def _compile_obj(src, obj, defs, verbose):
    _run([hipcc_path()] + FLAGS + defs + ["-c", src, "-o", obj + ".tmp"], verbose)
    os.replace(obj + ".tmp", obj)
    return obj


def build(force=False, verbose=True, jobs=None):
    want = tree_stamps()
    if force or embedded_stamps(MOTION_OUT).get("motion") != want["motion"]:
        _compile_shared(MOTION_SRC, MOTION_OUT, _stamp_def(want["motion"]), verbose)
    os.makedirs(OBJ, exist_ok=True)
    work = []
    rib_o = os.path.join(OBJ, "rib.o")
    if force or embedded_stamps(rib_o).get("lib") != want["lib"]:
        work.append((SRC, rib_o, _stamp_def(want["lib"]) + ['-DRIB_SHARD_STAMP="%s"' % want["shard0"]]))
    objs = [rib_o]
    for s in range(NSECTIONS):
        o = os.path.join(OBJ, "igemm_shard_%d.o" % s)
        objs.append(o)
        if force or embedded_stamps(o).get("shard%d" % s) != want["shard%d" % s]:
            work.append((SHARD_SRC, o, _stamp_def(want["shard%d" % s]) + ["-DRIB_SECTION=%d" % s, "-DRIB_ON_%d=RIB_KEEP" % s]))
    if work:
        work.sort(key=lambda w: w[0] != SRC)          # rib.o first: it is the longest single compile
        jobs = jobs or max(1, min(len(work), os.cpu_count() or 1))
        with ThreadPoolExecutor(jobs) as ex:
            for f in [ex.submit(_compile_obj, s, o, d, verbose) for s, o, d in work]:
                f.result()
    if work or needs_build(OUT):
        _run([hipcc_path()] + LINK_FLAGS + objs + ["-o", OUT + ".tmp"], verbose)
        os.replace(OUT + ".tmp", OUT)
    bad = check()
    if bad:
        raise RuntimeError("build stamps of the library differ from the tree's: %s" % bad)
    return OUT


def check():
    """[(tag, tree stamp, library stamp)] of every mismatch between the tracked sources and the built libraries."""
    want = tree_stamps()
    have = dict(embedded_stamps(OUT))
    have.update({k: v for k, v in embedded_stamps(MOTION_OUT).items() if k == "motion"})
    return [(t, want[t], have.get(t)) for t in sorted(want) if have.get(t) != want[t]]


if __name__ == "__main__":
    if "--check" in sys.argv:
        bad = check()
        for t, w in sorted(tree_stamps().items()):
            print("%-8s tree %s" % (t, w))
        for t, w, h in bad:
            print("MISMATCH %-8s tree %s library %s" % (t, w, h))
        sys.exit(1 if bad else 0)
    j = int(sys.argv[sys.argv.index("--jobs") + 1]) if "--jobs" in sys.argv else None
    build(force="--force" in sys.argv, jobs=j)
    print(OUT, os.path.getsize(OUT), "bytes, stamp", embedded_stamps(OUT).get("lib"))
