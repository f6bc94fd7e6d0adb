#!/usr/bin/env python3
"""A/B timing of several builds of librib.so inside ONE process / gpurun call (box-to-box variance is 2-5 %).
Uses only the entry points every build since round 1 exports (create, set_tensor, finalize, set_choice, forward, blend).

    python3 tools/ab_lib.py --tuning render-in-between_amd/tuning_gfx950.json lib_a.so lib_b.so ...
(tools/tuning_r01.json is round 1's table: the common ground when a build from round 1 is one of the contestants)
"""
import argparse, ctypes as C, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import render_in_between_amd as rib
from render_in_between_amd import synth, _native


def load(path):
    L = C.CDLL(path)
    L.rib_create.argtypes = [C.POINTER(_native.RibConfig), C.c_int, C.POINTER(C.c_void_p)]
    L.rib_set_tensor.argtypes = [C.c_void_p, C.c_char_p, C.c_void_p, C.c_int, C.POINTER(C.c_int64)]
    L.rib_finalize_weights.argtypes = [C.c_void_p]
    L.rib_workspace_bytes.restype = C.c_size_t
    L.rib_workspace_bytes.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int]
    L.rib_forward.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int] + [C.c_void_p] * 6 + [C.c_size_t, C.c_void_p]
    L.rib_blend.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int] + [C.c_void_p] * 5
    L.rib_set_choice.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_char_p, C.c_int, C.c_int]
    L.rib_variant_info.argtypes = [C.c_int, C.POINTER(C.c_int)]
    L.rib_last_error.restype = C.c_char_p
    L.rib_last_error.argtypes = [C.c_void_p]
    return L


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("libs", nargs="+")
    ap.add_argument("--tuning", default=None)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--size", type=int, default=512)
    a = ap.parse_args()
    cfg = rib.hsm_gen_config(); spec = rib.GenSpec.from_cfg(cfg)
    sd = synth.make_state_dict(spec, 0)
    B, H, W = 1, a.size, a.size
    label, fake, prev = [t.cuda() for t in synth.make_inputs(spec, B, H, W, 0)]
    img = torch.empty((B, 3, H, W), device="cuda"); mask = torch.empty((B, 1, H, W), device="cuda"); fuse = torch.empty_like(img)
    table = json.load(open(a.tuning)).get("%d,%d,%d" % (B, H, W), {}) if a.tuning else {}
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    handles = []
    for path in a.libs:
        L = load(os.path.abspath(path))
        c = _native.RibConfig(**{n: getattr(spec, n) for n, _ in _native.RibConfig._fields_})
        h = C.c_void_p(); assert L.rib_create(C.byref(c), 0, C.byref(h)) == 0
        for k, v in sd.items():
            t = v.contiguous(); d = (C.c_int64 * t.dim())(*t.shape)
            assert L.rib_set_tensor(h, k.encode(), C.c_void_p(t.data_ptr()), t.dim(), d) == 0
        assert L.rib_finalize_weights(h) == 0
        g = (C.c_int * 12)(); geoms = {}
        for i in range(L.rib_num_variants()):
            if L.rib_variant_info(i, g) == 0: geoms[tuple(g)] = i
        n = 0
        for op, ch in table.items():
            idx = geoms.get(tuple(ch[:10]) + (int(ch[11]) if len(ch) > 11 else 1, int(ch[12]) if len(ch) > 12 else 1))
            if idx is not None and L.rib_set_choice(h, B, H, W, op.encode(), idx, int(ch[10])) == 0: n += 1
        nb = L.rib_workspace_bytes(h, B, H, W)
        if nb == 0:      # a stale choice: fall back to the cost model for everything
            for op in table: L.rib_set_choice(h, B, H, W, op.encode(), -1, 1)
            nb = L.rib_workspace_bytes(h, B, H, W); n = 0
        ws = torch.empty(nb, dtype=torch.uint8, device="cuda")
        handles.append((path, L, h, ws, n))

    def step(L, h, ws):
        rc = L.rib_forward(h, B, H, W, label.data_ptr(), fake.data_ptr(), prev.data_ptr(), img.data_ptr(), mask.data_ptr(), ws.data_ptr(), ws.numel(), st)
        assert rc == 0, L.rib_last_error(h)
        L.rib_blend(h, B, 3, H, W, img.data_ptr(), mask.data_ptr(), fake.data_ptr(), fuse.data_ptr(), st)
    for r in range(a.rounds):
        for path, L, h, ws, n in handles:
            for _ in range(20): step(L, h, ws)
            torch.cuda.synchronize()
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(a.steps): step(L, h, ws)
            e1.record(); torch.cuda.synchronize()
            print("round %d  %-50s tuned ops %3d  ws %6.1f MB  %.4f ms/frame" % (r, os.path.basename(path), n, ws.numel() / 1e6, e0.elapsed_time(e1) / a.steps), flush=True)


if __name__ == "__main__":
    main()
