"""CPU suite for the native library: the C ABI loads and exports every symbol
of include/rib.h, and the host-side logic (layer inventory, spectral-norm fold,
filter re-layout, launch plans) agrees with the build's Python spec and with
the oracle.  No compute entry point is called (there is no GPU here); the
handle is created host-only (device = -1)."""
import ctypes as C
import os
import re

import numpy as np
import pytest
import torch

import render_in_between_amd as rib
from render_in_between_amd import _native, synth
from oracle import generator_ref

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FIELDS = [n for n, _ in _native.RibConfig._fields_]
MID_CFG = dict(num_filters=16, max_num_filters=64,
               mask=dict(num_filters=32, max_num_filters=64),
               embed=dict(num_filters=32, max_num_filters=64))


@pytest.fixture(scope="module")
def lib():
    if not os.path.exists(_native.LIB_PATH):
        from importlib import util
        spec = util.spec_from_file_location("rib_build", os.path.join(os.path.dirname(_native.LIB_PATH), "build.py"))
        mod = util.module_from_spec(spec); spec.loader.exec_module(mod)
        mod.build()
    return _native.lib()


def host_handle(lib, cfg):
    spec = rib.GenSpec.from_cfg(cfg)
    c = _native.RibConfig(**{n: getattr(spec, n) for n in FIELDS})
    h = C.c_void_p()
    rc = lib.rib_create(C.byref(c), -1, C.byref(h))
    assert rc == 0, lib.rib_last_error(None)
    return spec, h


def test_header_symbols_all_exported_and_bound(lib):
    hdr = open(os.path.join(ROOT, "include", "rib.h")).read()
    declared = set(re.findall(r"\b(rib_[a-z_]+)\s*\(", hdr))
    assert len(declared) >= 25
    for name in declared:
        assert hasattr(lib, name), "librib.so does not export %s" % name
        assert name in _native.SIGNATURES, "no ctypes signature for %s" % name
    assert set(_native.SIGNATURES) == declared
    # the struct the binding passes matches the header field for field
    fields = re.findall(r"int32_t\s+(\w+);", hdr)
    assert fields == FIELDS


def _build_module():
    from importlib import util
    spec = util.spec_from_file_location("rib_build", os.path.join(os.path.dirname(_native.LIB_PATH), "build.py"))
    mod = util.module_from_spec(spec); spec.loader.exec_module(mod)
    return mod


def test_library_is_what_the_tracked_sources_build(lib):
    """the measured binary must be the one the tree compiles to.  Every object of librib.so carries
    the content hash of the sources it was compiled from (csrc/build.py); the hashes of the tree as it is now, the hashes
    found in the .so file and the ones rib_build_info() reports at run time must all agree, for every object (rib.o and the shard objects)."""
    b = _build_module()
    want = b.tree_stamps()
    assert b.check() == [], "librib.so / libribmotion.so were not built from this tree: run csrc/build.py"
    have = b.embedded_stamps(b.OUT)
    assert {t: have.get(t) for t in want if t != "motion"} == {t: want[t] for t in want if t != "motion"}
    info = _native.build_info()
    assert info["consistent"] and info["stamp"] == want["lib"]
    assert info["shards"] == [want["shard%d" % s] for s in range(b.NSECTIONS)]
    assert info["variants"] == lib.rib_num_variants()
    # a changed source changes the stamp (content, not mtime)
    assert b.stamp_of(b.SHARD_DEPS, extra=("x",)) != want["shard0"]


def test_library_exports_only_the_declared_c_abi(lib):
    """No un-prefixed helper leaks out of librib.so: the defined dynamic function symbols are the
    rib_* entry points of include/rib.h plus mangled C++ / HIP registration symbols."""
    import subprocess
    out = subprocess.run(["nm", "-D", "--defined-only", _native.LIB_PATH], capture_output=True, text=True, check=True).stdout
    plain = [l.split()[-1] for l in out.splitlines() if l.split()[1] in "TtWw" and not l.split()[-1].startswith(("_Z", "__", "_init", "_fini"))]
    stray = sorted(n for n in plain if not n.startswith("rib_"))
    assert stray == [], stray
    hdr = open(os.path.join(ROOT, "include", "rib.h")).read()
    declared = set(re.findall(r"\b(rib_[a-z_]+)\s*\(", hdr))
    assert set(plain) <= declared, sorted(set(plain) - declared)


def test_native_inventory_matches_python_spec(lib):
    spec, h = host_handle(lib, rib.hsm_gen_config())
    want = dict(rib.state_dict_spec(spec))
    name = C.c_char_p(); ndim = C.c_int(); dims = (C.c_int64 * 4)(); used = C.c_int()
    got = {}
    for i in range(lib.rib_num_tensors(h)):
        assert lib.rib_tensor_info(h, i, C.byref(name), C.byref(ndim), dims, C.byref(used)) == 0
        got[name.value.decode()] = (tuple(dims[j] for j in range(ndim.value)), bool(used.value))
    assert len(got) == 372 and set(got) == set(want)
    for k, (shape, used_) in got.items():
        assert shape == want[k], k
        assert used_ == (not k.startswith(("label_embedding.", "conv_mask."))), k
    lib.rib_destroy(h)


def test_strict_load_errors(lib):
    spec, h = host_handle(lib, rib.hsm_gen_config(**MID_CFG))
    assert lib.rib_finalize_weights(h) == -4                      # RIB_ERR_MISSING
    assert b"missing key" in lib.rib_last_error(h)
    x = np.zeros(4, np.float32)
    d = (C.c_int64 * 1)(4)
    assert lib.rib_set_tensor(h, b"no.such.tensor", x.ctypes.data_as(C.c_void_p), 1, d) == -1
    assert b"unexpected key" in lib.rib_last_error(h)
    assert lib.rib_set_tensor(h, b"down_first.layers.conv.bias", x.ctypes.data_as(C.c_void_p), 1, d) == -1
    assert b"size mismatch" in lib.rib_last_error(h)
    lib.rib_destroy(h)


@pytest.mark.parametrize("cfgname", ["mid", "full"])
def test_fold_and_layout_match_oracle(lib, cfgname):
    cfg = rib.hsm_gen_config(**MID_CFG) if cfgname == "mid" else rib.hsm_gen_config()
    spec, h = host_handle(lib, cfg)
    sd = synth.make_state_dict(spec, 21)
    for k, v in sd.items():
        t = v.contiguous()
        d = (C.c_int64 * t.dim())(*t.shape)
        assert lib.rib_set_tensor(h, k.encode(), C.c_void_p(t.data_ptr()), t.dim(), d) == 0, lib.rib_last_error(h)
    assert lib.rib_finalize_weights(h) == 0, lib.rib_last_error(h)
    convs = [c for c in rib.conv_inventory(spec) if c.used]
    if cfgname == "full":   # a sample keeps the full-size case quick
        convs = [c for c in convs if c.name in ("ref_embedding.down_3", "down_first", "up_4.conv_block_s",
                                                "res_1.conv_block_1", "down_0.conv_block_0", "down_0.conv_block_s",
                                                "flow_network_temp.res_flow.0.conv_block_0", "conv_img")]
    for c in convs:
        w_ref, b_ref = generator_ref.conv_weight(sd, c.name)
        w = np.empty(tuple(w_ref.shape), np.float32); b = np.empty(c.cout, np.float32)
        assert lib.rib_debug_conv_weight(h, c.name.encode(), w.ctypes.data_as(C.c_void_p), b.ctypes.data_as(C.c_void_p)) == 0
        scale = float(w_ref.abs().max())
        assert np.abs(w - w_ref.numpy()).max() <= 2e-6 * scale, c.name      # fp64 vs fp32 sigma
        assert np.array_equal(b, b_ref.numpy()), c.name
        if c.spade_cond:
            p = c.name + ".layers.norm.mlps.0.0.layers.conv"
            ws = np.empty((2 * c.cin, c.spade_cond), np.float32); bs = np.empty(2 * c.cin, np.float32)
            assert lib.rib_debug_spade_weight(h, c.name.encode(), ws.ctypes.data_as(C.c_void_p), bs.ctypes.data_as(C.c_void_p)) == 0
            assert np.array_equal(ws, sd[p + ".weight"].numpy().reshape(2 * c.cin, c.spade_cond)), c.name
            assert np.array_equal(bs, sd[p + ".bias"].numpy()), c.name
    lib.rib_destroy(h)


def test_partial_reload_after_finalize_folds_old_and_new_tensors(lib):
    """load_state_dict(subset, strict=False) after a full load: rib_finalize_weights must fold the new tensors together
    with the ones it already holds (the host copies stay after a finalize), not read freed storage."""
    cfg = rib.hsm_gen_config(**MID_CFG)
    spec, h = host_handle(lib, cfg)
    sd = synth.make_state_dict(spec, 21)

    def put(k, v):
        t = v.contiguous()
        d = (C.c_int64 * t.dim())(*t.shape)
        assert lib.rib_set_tensor(h, k.encode(), C.c_void_p(t.data_ptr()), t.dim(), d) == 0, lib.rib_last_error(h)

    for k, v in sd.items():
        put(k, v)
    assert lib.rib_finalize_weights(h) == 0, lib.rib_last_error(h)
    # second, partial load: one spectral-norm conv gets a new weight_orig (u, v, bias and every other layer stay)
    name = "down_1.conv_block_0"
    sd2 = dict(sd)
    sd2[name + ".layers.conv.weight_orig"] = sd[name + ".layers.conv.weight_orig"] * 1.5 + 0.01
    put(name + ".layers.conv.weight_orig", sd2[name + ".layers.conv.weight_orig"])
    assert lib.rib_finalize_weights(h) == 0, lib.rib_last_error(h)
    for cname, ref_sd in ((name, sd2), ("down_1.conv_block_1", sd), ("conv_img", sd)):
        w_ref, b_ref = generator_ref.conv_weight(ref_sd, cname)
        w = np.empty(tuple(w_ref.shape), np.float32); b = np.empty(w_ref.shape[0], np.float32)
        assert lib.rib_debug_conv_weight(h, cname.encode(), w.ctypes.data_as(C.c_void_p), b.ctypes.data_as(C.c_void_p)) == 0
        assert np.abs(w - w_ref.numpy()).max() <= 2e-6 * float(w_ref.abs().max()), cname
        assert np.array_equal(b, b_ref.numpy()), cname
    lib.rib_destroy(h)


def test_precision_mode_switch_refolds_the_weights(lib):
    """rib_set_compute_dtype changes the blob layout (bf16: 16-channel minimum, bf16 filter copies).  A handle that still holds the state-dict tensors re-folds by itself; the fp32 section still undoes to the
    oracle's fold; plans of every mode build at odd sizes; an unknown mode is refused."""
    cfg = rib.hsm_gen_config(**MID_CFG)
    spec, h = host_handle(lib, cfg)
    sd = synth.make_state_dict(spec, 21)
    for k, v in sd.items():
        t = v.contiguous()
        d = (C.c_int64 * t.dim())(*t.shape)
        assert lib.rib_set_tensor(h, k.encode(), C.c_void_p(t.data_ptr()), t.dim(), d) == 0
    assert lib.rib_finalize_weights(h) == 0
    n32 = lib.rib_weights_bytes(h)
    sizes = {}
    for mode in (1, 3, 0):       # bf16, half, fp32
        assert lib.rib_set_compute_dtype(h, mode) == 0, lib.rib_last_error(h)
        sizes[mode] = lib.rib_weights_bytes(h)
        for cname in ("ref_embedding.conv_first", "down_1.conv_block_0", "conv_img"):
            w_ref, b_ref = generator_ref.conv_weight(sd, cname)
            w = np.empty(tuple(w_ref.shape), np.float32); b = np.empty(w_ref.shape[0], np.float32)
            assert lib.rib_debug_conv_weight(h, cname.encode(), w.ctypes.data_as(C.c_void_p), b.ctypes.data_as(C.c_void_p)) == 0, lib.rib_last_error(h)
            assert np.abs(w - w_ref.numpy()).max() <= 2e-6 * float(w_ref.abs().max()), (mode, cname)
        for (B, H, W) in ((1, 64, 64), (2, 48, 80), (1, 16, 16)):
            assert lib.rib_workspace_bytes(h, B, H, W) > 0, (mode, lib.rib_last_error(h))
    # (fp32 carries the Winograd-domain filters of the deep 3x3 layers; bf16 carries bf16 copies instead)
    assert sizes[0] == n32 and sizes[1] != n32 and sizes[3] == sizes[1]          # (the two 16-bit modes share a layout)
    assert lib.rib_set_compute_dtype(h, 7) != 0 and lib.rib_set_compute_dtype(h, 2) != 0      # (2 was round 2's retired f32x3 mode)
    lib.rib_destroy(h)


def test_half_mode_refuses_filters_beyond_its_range_and_products_setting_is_checked(lib):
    """IEEE half ends at 65504: rib_finalize_weights in RIB_DTYPE_F16 refuses a checkpoint whose FOLDED filter leaves that range
    (it would become an infinity in the 16-bit copy and a NaN frame later) and names the layer; bf16 and fp32 take the same
    tensors.  rib_set_products accepts exactly its two settings."""
    cfg = rib.hsm_gen_config(**MID_CFG)
    spec, h = host_handle(lib, cfg)
    sd = synth.make_state_dict(spec, 5)
    sd["conv_img.layers.conv.weight"] = sd["conv_img.layers.conv.weight"] * 1e6          # (no spectral norm on this layer: folded = stored)
    for k, v in sd.items():
        t = v.contiguous()
        d = (C.c_int64 * t.dim())(*t.shape)
        assert lib.rib_set_tensor(h, k.encode(), C.c_void_p(t.data_ptr()), t.dim(), d) == 0
    assert lib.rib_finalize_weights(h) == 0                                                  # fp32: fine
    assert lib.rib_set_compute_dtype(h, 1) == 0                                              # bf16 (fp32's exponent range): fine
    assert lib.rib_set_compute_dtype(h, 3) != 0                                              # half: refused, by name
    msg = lib.rib_last_error(h).decode()
    assert "conv_img" in msg and "65504" in msg and "RIB_DTYPE_BF16" in msg, msg
    assert lib.rib_set_compute_dtype(h, 0) == 0 and lib.rib_workspace_bytes(h, 1, 64, 64) > 0   # the handle is still usable in another mode
    assert lib.rib_set_products(h, 1) == 0 and lib.rib_set_products(h, 0) == 0
    assert lib.rib_set_products(h, 2) != 0 and lib.rib_set_products(None, 0) != 0
    lib.rib_destroy(h)


def test_plan_flops_and_shape_rules(lib):
    spec, h = host_handle(lib, rib.hsm_gen_config())
    fl = (C.c_double * len(_native.KC_NAMES))()
    for (B, H, W) in [(1, 512, 512), (1, 320, 480), (2, 64, 64), (4, 1024, 1024)]:
        assert lib.rib_forward_flops(h, B, H, W, fl) == 0, lib.rib_last_error(h)
        assert abs(sum(fl) - B * rib.conv_flops(spec, H, W)) < 1e-6 * sum(fl)
        assert lib.rib_workspace_bytes(h, B, H, W) > 0
    assert abs(sum(fl) / 4 / 1e9 - 926.4) < 0.1                     # SURVEY §8d, 1024^2 per sample
    assert lib.rib_num_launches(h, 1, 512, 512) > 100
    # sizes not divisible by 16 are rejected (the reference itself fails on them)
    assert lib.rib_workspace_bytes(h, 1, 250, 250) == 0
    assert b"multiples of 16" in lib.rib_last_error(h)
    # every launch of the plan is describable and every grid is non-empty
    buf = C.create_string_buffer(512)
    n = lib.rib_num_launches(h, 1, 512, 512)
    names = []
    for i in range(n):
        assert lib.rib_debug_launch_info(h, 1, 512, 512, i, buf, 512) == 0
        name, kclass, grid, _, _, _ = buf.value.decode().split("|")
        assert all(int(g) > 0 for g in grid.split(","))
        names.append(name)
    # (the deep 3x3 layers run in the Winograd domain at this size: transform, batched GEMM, transform)
    for must in ("down_0.0.spade", "res_1.conv_block_1.wino_in", "res_1.conv_block_1.wino", "res_1.conv_block_1.wino_out", "flow_network_temp.res_flow.2.conv_block_0.wino4", "up_0.conv_block_1", "conv_img",
                 "flow_network_temp.res_flow.3.join", "flow_network_temp.conv_mask.0"):
        assert must in names
    lib.rib_destroy(h)


def test_unsupported_configs_fail_loudly(lib):
    spec = rib.GenSpec.from_cfg(rib.hsm_gen_config(embed=dict(num_filters=24)))
    c = _native.RibConfig(**{n: getattr(spec, n) for n in FIELDS})
    h = C.c_void_p()
    assert lib.rib_create(C.byref(c), -1, C.byref(h)) == -2
    assert b"multiple of 32" in lib.rib_last_error(None)


def test_the_product_package_never_touches_the_oracle():
    """oracle/ is test infrastructure: nothing under render-in-between_amd/ (the product: host mirror, driver, CLIs, C++ / HIP
    sources) and neither bench.py's timed path nor inference.py may import, open or name it.  bench.py's one import is the
    cpu_baseline leg, after the timed region; importing the whole product package must not pull the oracle in either."""
    import subprocess, sys
    pkg = os.path.join(ROOT, "render-in-between_amd")
    hits = []
    for dirpath, _, files in os.walk(pkg):
        if "__pycache__" in dirpath or os.path.join("csrc", "build") in dirpath:
            continue
        for f in files:
            if not f.endswith((".py", ".hip", ".h", ".def", ".yaml")):
                continue
            text = open(os.path.join(dirpath, f), errors="replace").read()
            for m in re.finditer(r"^\s*(from|import)\s+oracle\b|['\"/]oracle/", text, flags=re.M):
                hits.append((os.path.relpath(os.path.join(dirpath, f), ROOT), m.group(0).strip()))
    assert hits == [], hits
    bench = open(os.path.join(ROOT, "bench.py")).read()
    imports = [m.start() for m in re.finditer(r"^\s*from oracle import", bench, flags=re.M)]
    assert len(imports) == 1 and imports[0] > bench.index("dt = time.perf_counter() - t0"), "bench.py: the oracle may only appear in the cpu_baseline leg, after the timed region"
    code = ("import sys; sys.path.insert(0, %r); import render_in_between_amd as rib; "
            "from render_in_between_amd import evaluator, distributed, rasterise, resize, tuning, io_worker, config, spec, synth; "
            "from render_in_between_amd.motion import model, pose_io, spec as mspec; "
            "assert not [m for m in sys.modules if m == 'oracle' or m.startswith('oracle.')], 'the product imported the oracle'" % ROOT)
    subprocess.run([sys.executable, "-c", code], check=True)


def test_generator_refuses_to_run_without_gpu():
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(RuntimeError, match="no CPU path"):
        rib.Generator(rib.hsm_gen_config())


def test_every_tuned_choice_names_an_existing_kernel_variant(lib):
    """tuning.apply() silently falls back to the cost model for a geometry that is no longer in the variant
    table: after editing the table, stale entries would go unnoticed.  Every entry must resolve."""
    import json
    from render_in_between_amd import tuning
    g12 = (C.c_int * 12)()
    geoms = set()
    for i in range(lib.rib_num_variants()):
        if lib.rib_variant_info(i, g12) == 0:            # fp32 variants (this table is measured in fp32)
            geoms.add(tuple(g12))
    table = tuning.load()
    assert "1,512,512" in table and len(table["1,512,512"]) >= 40      # (round 3: 59 tunable launches at 512x512; those whose default is the measured winner have no entry)
    stale = []
    for shape, entry in table.items():
        for op, ch in entry.items():
            kw = int(ch[11]) if len(ch) > 11 else 1
            tb = int(ch[12]) if len(ch) > 12 else 1
            if tuple(ch[:10]) + (kw, tb) not in geoms:
                stale.append((shape, op, ch))
            assert int(ch[10]) >= 1
    assert not stale, stale[:5]
    # the bf16-storage mode has a table of its own, measured on its own kernels
    for dtype in ("bf16", "f16"):          # (half runs the bf16 geometries with another element type: same table)
        geoms = set()
        for i in range(lib.rib_num_variants()):
            if lib.rib_variant_info(i, g12) == tuning.PREC[dtype]:
                geoms.add(tuple(g12))
        assert len(geoms) >= 40, (dtype, len(geoms))
        table = tuning.load(dtype=dtype)
        assert "1,512,512" in table and len(table["1,512,512"]) >= 50, dtype
        stale = [(shape, op) for shape, entry in table.items() for op, ch in entry.items()
                 if tuple(ch[:10]) + (int(ch[11]), int(ch[12])) not in geoms]
        assert not stale, (dtype, stale[:5])


def test_every_kernel_variant_of_the_library_is_used_by_some_plan(lib):
    """Round 4 pruned csrc/variants.def to the entries some launch plan uses (tools/prune_variants.py): an entry that no plan of
    the sweep - both test configurations, the three precision modes, every tabled shape and 50 others, with and without the
    tables - ever launches is dead weight in a 14 MB library and 2+ minutes of build (k_gemm_dma's tiles, FRW = 0, are chosen
    per GEMM at plan time and do not show up in the launch strings this check reads)."""
    import sys
    sys.path.insert(0, ROOT)
    from tools.variant_usage import usage
    info, used = usage()
    unused = [tuple(info[i][1]) + (info[i][0],) for i in range(len(info)) if i not in used and info[i][1][0] != 0]
    # Candidates added for a tuning run are unused until a table picks them: they are listed HERE, by geometry (+ precision), so
    # that widening the autotuner's search space is an explicit edit and anything else that falls out of use fails the test
    # (the limit had been relaxed to 16 with a warning - a dozen orphans would have passed silently).
    EXPECTED_UNUSED = set()          # currently none: round 6's 32x16-tile candidates lost and were removed again
    stray = [g for g in unused if g not in EXPECTED_UNUSED]
    assert not stray, "kernel variants launched by no plan of the sweep (prune them or list them as tuning candidates): %s" % stray
    assert EXPECTED_UNUSED <= set(unused), "listed as unused but launched: %s" % sorted(EXPECTED_UNUSED - set(unused))
