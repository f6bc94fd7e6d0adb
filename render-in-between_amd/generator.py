"""Drop-in host-side mirror of the reference ``Generator`` object protocol
(PGNR/models/generator.py:35-234 as used by PGNR/models/trainer.py:61,67 and
PGNR/models/evaluator.py:170,255):

    net_G = Generator(cfg.gen)            # same ctor argument
    net_G.load_state_dict(state_dict)     # same 372-tensor checkpoint, strict
    net_G.eval()
    img, mask = net_G(label, label_prev, img_fake, img_prev)

All compute happens in hand-written HIP kernels behind the C ABI of
include/rib.h; this class only validates arguments, owns the workspace tensor
and passes device pointers.  No CPU fallback exists: constructing a Generator
without a GPU or without the built library raises.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, Optional

import torch

from . import _native
from .config import GenSpec


def _ptr(t: Optional[torch.Tensor]):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


class Generator:
    def __init__(self, gen_cfg, device=None, use_tuning=True, compute_dtype="f32", products="f32"):
        """compute_dtype: 'f32' (exact-fp32 matrix cores; the reference's arithmetic), 'bf16' (bf16 storage and
        matrix-core operands, fp32 accumulate / statistics; BASELINE config 3) or 'f16' (the same 16-bit kernels
        with IEEE half elements: ~10x closer to fp32 than bf16 at the same speed).
        products (fp32 mode only, opt-in): 'f32' = exact-fp32 products everywhere (default); 'bf16x3' = the plain GEMMs of
        the frame form every product from six bf16 matrix-core products of three-way split operands (include/rib.h,
        rib_set_products): fp32-grade, not the reference's arithmetic."""
        if compute_dtype not in ("f32", "bf16", "f16"):
            raise ValueError("compute_dtype must be 'f32', 'bf16' or 'f16'")
        if products not in ("f32", "bf16x3"):
            raise ValueError("products must be 'f32' or 'bf16x3'")
        if products != "f32" and compute_dtype != "f32":
            raise ValueError("products='bf16x3' is an option of the fp32 mode")
        self.compute_dtype = compute_dtype
        self.products = products
        self.spec = GenSpec.from_cfg(gen_cfg)
        self._tuning = None
        self._use_tuning = use_tuning
        self.gen_cfg = gen_cfg
        if not torch.cuda.is_available():
            raise RuntimeError("render_in_between_amd.Generator needs a ROCm GPU (MI355X); "
                               "there is no CPU path")
        if device is None:
            device = torch.device("cuda", torch.cuda.current_device())
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("Generator device must be a GPU, got %s" % (self.device,))
        if self.device.index is None:
            self.device = torch.device("cuda", torch.cuda.current_device())
        self._lib = _native.lib()
        s = self.spec
        cfg = _native.RibConfig(
            label_nc=s.label_nc, image_nc=s.image_nc, num_filters=s.num_filters,
            max_num_filters=s.max_num_filters, num_layers=s.num_layers,
            num_down_img=s.num_down_img, emb_filters=s.emb_filters,
            emb_max_filters=s.emb_max_filters, emb_down=s.emb_down,
            mask_filters=s.mask_filters, mask_max_filters=s.mask_max_filters,
            mask_down=s.mask_down, mask_res_blocks=s.mask_res_blocks)
        h = C.c_void_p()
        rc = self._lib.rib_create(C.byref(cfg), self.device.index, C.byref(h))
        if rc != 0:
            msg = self._lib.rib_last_error(None)
            raise (NotImplementedError if rc == -2 else _native.RibError)(
                *(("rib_create: " + msg.decode(),) if rc == -2 else (rc, msg.decode())))
        self._h = h
        _native.check(h, self._lib.rib_set_compute_dtype(h, {"f32": 0, "bf16": 1, "f16": 3}[compute_dtype]))
        _native.check(h, self._lib.rib_set_products(h, {"f32": 0, "bf16x3": 1}[products]))
        self._ws: Dict[tuple, torch.Tensor] = {}
        self._plan_batch = 0
        self._tuned: Dict[tuple, int] = {}      # (B,H,W) -> launches whose variant came from the measured table
        self._applied: Dict[tuple, int] = {}    # (TB,H,W) whose table entries have been pinned on the handle -> how many
        self.training = False
        self._graph_replay = bool(int(__import__("os").environ.get("RIB_GRAPH", "0") or 0))
        self.weights_version = 0        # bumped by load_state_dict / import_weights (Evaluator's lane clones follow it)
        self._warned_copy = False

    def __del__(self):
        h = getattr(self, "_h", None)
        if h:
            self._lib.rib_destroy(h)
            self._h = None

    # ---- nn.Module-protocol no-ops the reference driver calls ----------------------------
    def eval(self):
        self.training = False
        return self

    def train(self, mode=True):
        if mode:
            raise NotImplementedError("the MI355X path is inference-only")
        return self.eval()

    def to(self, device=None, *a, **k):
        if device is not None and torch.device(device).type == "cuda":
            idx = torch.device(device).index
            if idx is not None and idx != self.device.index:
                raise RuntimeError("a Generator handle is bound to one GPU; construct it on %s" % device)
        return self

    def cuda(self, device=None):
        return self.to("cuda" if device is None else device)

    # ---- checkpoint surface ---------------------------------------------------------------
    def expected_tensors(self):
        """[(name, shape, used)] exactly as the native loader expects them."""
        out = []
        name = C.c_char_p(); ndim = C.c_int(); dims = (C.c_int64 * 4)(); used = C.c_int()
        for i in range(self._lib.rib_num_tensors(self._h)):
            _native.check(self._h, self._lib.rib_tensor_info(self._h, i, C.byref(name), C.byref(ndim), dims, C.byref(used)))
            out.append((name.value.decode(), tuple(dims[j] for j in range(ndim.value)), bool(used.value)))
        return out

    def load_state_dict(self, state_dict, strict=True):
        """Same contract as nn.Module.load_state_dict(strict=True) on the
        reference generator (PGNR/utils/utils.py:107-119): every reference key
        must be present with its shape; unknown keys are an error."""
        if "state_dict" in state_dict and not torch.is_tensor(state_dict["state_dict"]):
            state_dict = state_dict["state_dict"]                      # utils.py:115-116
        sd = {k.replace("module.", ""): v for k, v in state_dict.items()}   # utils.py:101-105
        expected = {n for n, _, _ in self.expected_tensors()}
        if strict:
            missing = sorted(expected - set(sd))
            unexpected = sorted(set(sd) - expected)
            if missing or unexpected:
                raise RuntimeError("Error(s) in loading state_dict for Generator:\n\tMissing key(s): %s\n\t"
                                   "Unexpected key(s): %s" % (missing[:8], unexpected[:8]))
        for k, v in sd.items():
            if k not in expected:
                continue
            t = v.detach().to("cpu", torch.float32).contiguous()
            dims = (C.c_int64 * max(t.dim(), 1))(*t.shape)
            _native.check(self._h, self._lib.rib_set_tensor(self._h, k.encode(), C.c_void_p(t.data_ptr()), t.dim(), dims))
        with torch.cuda.device(self.device):
            _native.check(self._h, self._lib.rib_finalize_weights(self._h))
        self.weights_version += 1
        if self.compute_dtype == "f16":
            self.assert_finite_probe()
        return self

    def assert_finite_probe(self, size=64, seed=0):
        """IEEE half ends at 65504.  rib_finalize_weights refuses a checkpoint whose FOLDED FILTERS leave that range; this
        checks the ACTIVATIONS: one small forward on full-range inputs (frames uniform in [-1, 1], label maps in their value
        ranges) must come out finite - the condition encoder is five convolutions with no normalisation between them, the one
        place where a trained checkpoint's larger filters can compound.  Raises FloatingPointError (the f16 mode must fail
        loudly, not render NaN frames); called by load_state_dict in the f16 mode."""
        m = 1 << max(self.spec.num_down_img + 1, self.spec.mask_down, self.spec.emb_down)
        hw = max(size // m, 1) * m
        g = torch.Generator().manual_seed(seed)
        label = torch.cat([torch.rand(1, 3, hw, hw, generator=g) * 2 - 1, torch.rand(1, self.spec.label_nc - 3, hw, hw, generator=g)], dim=1)
        fake, prev = torch.rand(1, self.spec.image_nc, hw, hw, generator=g) * 2 - 1, torch.rand(1, self.spec.image_nc, hw, hw, generator=g) * 2 - 1
        img, mask = self(label.to(self.device), None, fake.to(self.device), prev.to(self.device))
        if not (bool(torch.isfinite(img).all()) and bool(torch.isfinite(mask).all())):
            raise FloatingPointError("compute_dtype='%s': this checkpoint's activations leave the 16-bit format's range (non-finite output on "
                                     "a %dx%d probe frame); use compute_dtype='bf16' (same speed, fp32's range) or 'f32'" % (self.compute_dtype, hw, hw))
        return True

    # ---- multi-GPU weight hand-off (one RCCL broadcast of the folded blob) -----------------
    def export_weights(self) -> torch.Tensor:
        n = self._lib.rib_weights_bytes(self._h)
        buf = torch.empty(n // 4, dtype=torch.float32, device=self.device)
        _native.check(self._h, self._lib.rib_export_weights(self._h, _ptr(buf), n, self._stream()))
        return buf

    def clone(self) -> "Generator":
        """A second handle on the same device with the same folded weights (device-to-device copy
        of the blob): used to keep several independent segments in flight on separate HIP streams
        (a handle is single-stream)."""
        g = Generator(self.gen_cfg, device=self.device, use_tuning=self._use_tuning, compute_dtype=self.compute_dtype,
                      products=self.products)
        g.set_plan_batch(self._plan_batch)
        blob = self.export_weights()
        g.import_weights(blob)
        torch.cuda.current_stream(self.device).synchronize()
        return g

    def weights_numel(self) -> int:
        return self._lib.rib_weights_bytes(self._h) // 4

    def import_weights(self, buf: torch.Tensor):
        assert buf.is_cuda and buf.dtype == torch.float32 and buf.is_contiguous()
        _native.check(self._h, self._lib.rib_import_weights(self._h, _ptr(buf), buf.numel() * 4, self._stream()))
        self.weights_version += 1
        return self

    # ---- forward ----------------------------------------------------------------------------
    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def set_plan_batch(self, n):
        """Batch-invariant launch plans (rib_set_plan_batch, include/rib.h): with n > 0 every plan, whatever its batch, follows
        the kernel choices of batch n, so a sample's frames are bit-identical in every grouping (B = 1, a ragged group of 3,
        a full group); 0 = every batch its own choices (the default of a handle; fastest for a single call)."""
        n = max(0, int(n))
        if n != self._plan_batch:
            _native.check(self._h, self._lib.rib_set_plan_batch(self._h, n))
            self._plan_batch = n
            self._ws.clear()                     # the workspace a shape needs follows the plans
            self.__dict__.pop("_chain_ws", None)
            self.__dict__.pop("_chain_out", None)
        return self

    @property
    def plan_batch(self):
        return self._plan_batch

    def _workspace(self, B, H, W):
        key = (B, H, W)
        ws = self._ws.get(key)
        if ws is None:
            TB = self._plan_batch or B           # the batch whose measured choices this shape's plans follow
            if self._use_tuning:
                from . import tuning
                if self._tuning is None:
                    self._tuning = tuning.load(dtype=self.compute_dtype)
                if (TB, H, W) not in self._applied:      # once per followed shape: re-pinning drops the plans that follow it
                    self._applied[(TB, H, W)] = tuning.apply(self._lib, self._h, self._tuning, TB, H, W, dtype=self.compute_dtype)
                self._tuned[key] = self._applied[(TB, H, W)]
            n = self._lib.rib_workspace_bytes(self._h, B, H, W)
            if n == 0 and self._use_tuning and self._tuning.get("%d,%d,%d" % (TB, H, W)):
                # a stale tuning entry must never break the path: drop it and use the cost model
                for op in self._tuning["%d,%d,%d" % (TB, H, W)]:
                    self._lib.rib_set_choice(self._h, TB, H, W, op.encode(), -1, 1)
                self._applied[(TB, H, W)] = self._tuned[key] = 0
                # erasing the choices of (TB, H, W) rebuilds EVERY plan that follows TB, other batch sizes included: the
                # workspaces cached for them were sized under the old choices (rib_forward / rib_chain reject a workspace
                # smaller than the current plan needs, so a stale one could only fail loudly - but it need not fail at all)
                self._ws.clear()
                self.__dict__.pop("_chain_ws", None)
                self.__dict__.pop("_chain_out", None)
                n = self._lib.rib_workspace_bytes(self._h, B, H, W)
            if n == 0:
                _native.check(self._h, -1)
            ws = torch.empty(n, dtype=torch.uint8, device=self.device)
            self._ws[key] = ws
        return ws

    def _prep(self, t, ch, name, shape=None):
        if not torch.is_tensor(t) or t.dim() != 4 or t.shape[1] != ch:
            raise ValueError("%s must be a [B,%d,H,W] tensor, got %s" % (name, ch, tuple(getattr(t, "shape", ()))))
        if shape is not None and (t.shape[0], t.shape[2], t.shape[3]) != shape:
            raise ValueError("%s has shape %s, expected B,H,W = %s" % (name, tuple(t.shape), shape))
        if t.device != self.device or t.dtype != torch.float32 or not t.is_contiguous():
            if not self._warned_copy:
                # the reference driver keeps results['fuse'] on the CPU (evaluator.py:252): fed to this object unchanged
                # that is a host-to-device copy per frame; say so once instead of hiding it
                import warnings
                warnings.warn("Generator: %s arrived as %s/%s%s and is copied to %s float32 contiguous on every call; keep "
                              "the tensors on the GPU (or use Generator.chain) to avoid a per-frame copy"
                              % (name, t.device, str(t.dtype).replace("torch.", ""), "" if t.is_contiguous() else "/strided", self.device))
                self._warned_copy = True
            t = t.to(self.device, torch.float32)
        return t.contiguous()

    def __call__(self, label, label_prev, img_fake, img_prev):
        """img_final, mask = G(label, label_prev, img_fake, img_prev)
        (generator.py:181-234).  ``label_prev`` is accepted and ignored — the
        reference never reads it; it may be None."""
        s = self.spec
        label = self._prep(label, s.label_nc, "label")
        B, _, H, W = label.shape
        img_fake = self._prep(img_fake, s.image_nc, "img_fake", (B, H, W))
        img_prev = self._prep(img_prev, s.image_nc, "img_prev", (B, H, W))
        ws = self._workspace(B, H, W)
        img = torch.empty((B, s.image_nc, H, W), dtype=torch.float32, device=self.device)
        mask = torch.empty((B, 1, H, W), dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            _native.check(self._h, self._lib.rib_forward(
                self._h, B, H, W, _ptr(label), _ptr(img_fake), _ptr(img_prev), _ptr(img), _ptr(mask),
                _ptr(ws), ws.numel(), self._stream()))
        return img, mask

    forward = __call__

    def forward_blend(self, label, label_prev, img_fake, img_prev):
        """(img, mask, fuse): the forward plus the driver's blend fuse = img*mask + img_fake*(1-mask)
        (evaluator.py:256-258) in one call; the mask head's kernel writes the fused frame."""
        s = self.spec
        label = self._prep(label, s.label_nc, "label")
        B, _, H, W = label.shape
        img_fake = self._prep(img_fake, s.image_nc, "img_fake", (B, H, W))
        img_prev = self._prep(img_prev, s.image_nc, "img_prev", (B, H, W))
        ws = self._workspace(B, H, W)
        img = torch.empty((B, s.image_nc, H, W), dtype=torch.float32, device=self.device)
        fuse = torch.empty_like(img)
        mask = torch.empty((B, 1, H, W), dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            _native.check(self._h, self._lib.rib_forward_blend(
                self._h, B, H, W, _ptr(label), _ptr(img_fake), _ptr(img_prev), _ptr(img), _ptr(mask), _ptr(fuse),
                _ptr(ws), ws.numel(), self._stream()))
        return img, mask, fuse

    def chain(self, key_frame, labels, dains, want_all=True):
        """One autoregressive segment on device (evaluator.py:238-262):
        labels [T,B,label_nc,H,W], dains [T,B,image_nc,H,W], key_frame
        [B,image_nc,H,W] -> (imgs, masks, fuses); prev never leaves HBM."""
        s = self.spec
        T = labels.shape[0]
        B, H, W = labels.shape[1], labels.shape[3], labels.shape[4]
        labels = labels.to(self.device, torch.float32).contiguous()
        dains = dains.to(self.device, torch.float32).contiguous()
        key_frame = self._prep(key_frame, s.image_nc, "key_frame", (B, H, W))
        assert labels.shape == (T, B, s.label_nc, H, W) and dains.shape == (T, B, s.image_nc, H, W)
        ws = self._workspace(B, H, W)
        need = int(self._lib.rib_chain_workspace_bytes(self._h, T, B, H, W))      # + the batched label-only launches
        if need > ws.numel():
            # one grow-only buffer per batch size: the chunks of a segment differ in T (8, 8, 8, 7) and share it
            cache = self.__dict__.setdefault("_chain_ws", {})
            cws = cache.get((B, H, W))
            if cws is None or cws.numel() < need:
                cache.pop((B, H, W), None)
                cws = cache[(B, H, W)] = torch.empty(need, dtype=torch.uint8, device=self.device)
            ws = cws
        if self._graph_replay:
            # a replayed graph writes where it was captured: two alternating output sets per shape (the previous call's frames
            # stay valid while the next call runs - a chunked segment reads its `prev` from them), overwritten two calls later
            ring = self.__dict__.setdefault("_chain_out", {}).setdefault((T, B, H, W, bool(want_all)), {"n": 0, "sets": []})
            if len(ring["sets"]) < 2:
                fz = torch.empty((T, B, s.image_nc, H, W), dtype=torch.float32, device=self.device)
                ring["sets"].append((torch.empty_like(fz) if want_all else None,
                                     torch.empty((T, B, 1, H, W), dtype=torch.float32, device=self.device) if want_all else None, fz))
            imgs, masks, fuses = ring["sets"][ring["n"] % len(ring["sets"])] if len(ring["sets"]) == 2 else ring["sets"][-1]
            ring["n"] += 1
        else:
            fuses = torch.empty((T, B, s.image_nc, H, W), dtype=torch.float32, device=self.device)
            imgs = torch.empty_like(fuses) if want_all else None
            masks = torch.empty((T, B, 1, H, W), dtype=torch.float32, device=self.device) if want_all else None
        with torch.cuda.device(self.device):
            _native.check(self._h, self._lib.rib_chain(
                self._h, T, B, H, W, _ptr(key_frame), _ptr(labels), _ptr(dains), _ptr(imgs), _ptr(masks),
                _ptr(fuses), _ptr(ws), ws.numel(), self._stream()))
        return imgs, masks, fuses

    def set_graph_replay(self, on=True):
        """rib_chain as ONE HIP graph launch per (shape, tensors): see include/rib.h.  Also on with RIB_GRAPH=1."""
        _native.check(self._h, self._lib.rib_set_graph_replay(self._h, 1 if on else 0))
        self._graph_replay = bool(on)
        self.__dict__.pop("_chain_out", None)
        return self

    def graph_stats(self):
        cap, rep = C.c_int64(), C.c_int64()
        _native.check(self._h, self._lib.rib_graph_stats(self._h, C.byref(cap), C.byref(rep)))
        return {"captures": cap.value, "replays": rep.value}

    # ---- driver-side ops ----------------------------------------------------------------------
    def blend(self, img, mask, dain):
        img = img.to(self.device, torch.float32).contiguous(); mask = mask.to(self.device, torch.float32).contiguous()
        dain = dain.to(self.device, torch.float32).contiguous()
        B, Cc, H, W = img.shape
        out = torch.empty_like(img)
        _native.check(self._h, self._lib.rib_blend(self._h, B, Cc, H, W, _ptr(img), _ptr(mask), _ptr(dain), _ptr(out), self._stream()))
        return out

    def quantise(self, img):
        img = img.to(self.device, torch.float32).contiguous()
        B, Cc, H, W = img.shape
        out = torch.empty((B, H, W, Cc), dtype=torch.uint8, device=self.device)
        _native.check(self._h, self._lib.rib_quantise(self._h, B, Cc, H, W, _ptr(img), _ptr(out), self._stream()))
        return out

    def warp(self, img, flow):
        img = img.to(self.device, torch.float32).contiguous(); flow = flow.to(self.device, torch.float32).contiguous()
        B, Cc, H, W = img.shape
        assert flow.shape == (B, 2, H, W)
        out = torch.empty_like(img)
        _native.check(self._h, self._lib.rib_warp(self._h, B, Cc, H, W, _ptr(img), _ptr(flow), _ptr(out), self._stream()))
        return out

    def rasterise(self, strokes, peaks, weights, radius, height, width, colors=None, halfwidth=None):
        """Label maps of T frames drawn on the GPU (rib_rasterise): strokes [T, E] of
        rasterise.STROKE_DTYPE, peaks [T, P, 2] int32, weights [radius+1] fp64 (host arrays, see
        rasterise.py) -> [T, 3+P, H, W] fp32 CUDA tensor."""
        import numpy as np
        from . import rasterise as R
        strokes = np.ascontiguousarray(strokes, R.STROKE_DTYPE)
        peaks = np.ascontiguousarray(peaks, np.int32)
        weights = np.ascontiguousarray(weights, np.float64)
        colors = np.ascontiguousarray(R.POSE_COLORS if colors is None else colors, np.uint8)
        halfwidth = R.STROKE_HALFWIDTH if halfwidth is None else int(halfwidth)
        T, E = strokes.shape
        P = peaks.shape[1]
        assert peaks.shape == (T, P, 2) and colors.shape == (E, 3) and weights.shape == (radius + 1,)
        nbytes = self._lib.rib_rasterise_workspace_bytes(self._h, T, height, width, E, P, radius)
        with torch.cuda.device(self.device):
            ws = torch.empty(nbytes, dtype=torch.uint8, device=self.device)
            out = torch.empty((T, 3 + P, height, width), dtype=torch.float32, device=self.device)
            _native.check(self._h, self._lib.rib_rasterise(
                self._h, T, height, width, strokes.ctypes.data, E, colors.ctypes.data, halfwidth,
                peaks.ctypes.data, P, weights.ctypes.data, radius, _ptr(out), _ptr(ws), ws.numel(), self._stream()))
        return out

    # ---- introspection / measurement -----------------------------------------------------------
    def enable_taps(self, on=True):
        """Debug: keep every tapped intermediate intact until the end of a forward (buffers with disjoint lifetimes
        share workspace bytes otherwise).  Rebuilds the launch plans; call before the forward whose taps are read."""
        _native.check(self._h, self._lib.rib_set_debug_taps(self._h, 1 if on else 0))
        self._ws.clear()
        return self

    def read_taps(self, B, H, W):
        """Intermediate activations of the LAST forward at this shape (run after enable_taps()), as NCHW CPU tensors."""
        ws = self._workspace(B, H, W)
        out = {}
        name = C.c_char_p(); ch = C.c_int(); th = C.c_int(); tw = C.c_int()
        for i in range(self._lib.rib_num_taps(self._h, B, H, W)):
            _native.check(self._h, self._lib.rib_tap_info(self._h, B, H, W, i, C.byref(name), C.byref(ch), C.byref(th), C.byref(tw)))
            dst = torch.empty((B, ch.value, th.value, tw.value), dtype=torch.float32, device=self.device)
            _native.check(self._h, self._lib.rib_read_tap(self._h, B, H, W, i, _ptr(ws), _ptr(dst), self._stream()))
            out[name.value.decode()] = dst.cpu()
        return out

    def profile_begin(self, kernels=False):
        """kernels=False: one event in front of every launch (the classes add up to the profiled step, event cost included);
        kernels=True: a (start, stop) pair bound to every dispatch - the kernels' own execution times, as rocprofv3 reports them."""
        _native.check(self._h, (self._lib.rib_profile_begin_kernels if kernels else self._lib.rib_profile_begin)(self._h))

    def profile_collect(self):
        n = len(_native.KC_NAMES)
        launches = (C.c_int64 * n)(); ms = (C.c_double * n)()
        _native.check(self._h, self._lib.rib_profile_collect(self._h, launches, ms))
        return {k: {"launches": int(launches[i]), "ms": float(ms[i])} for i, k in enumerate(_native.KC_NAMES)}

    def forward_flops(self, B, H, W):
        n = len(_native.KC_NAMES)
        fl = (C.c_double * n)()
        _native.check(self._h, self._lib.rib_forward_flops(self._h, B, H, W, fl))
        return {k: float(fl[i]) for i, k in enumerate(_native.KC_NAMES)}

    def tuned_ops(self, B, H, W):
        """How many launches of this shape run a measured choice (0: the analytic cost model decides everything)."""
        self._workspace(B, H, W)
        return self._tuned.get((B, H, W), 0)

    def num_launches(self, B, H, W):
        return self._lib.rib_num_launches(self._h, B, H, W)

    def launch_info(self, B, H, W):
        """The launch plan of a shape: [{name, class (index into _native.KC_NAMES), grid, tile, flops}]."""
        self._workspace(B, H, W)                       # pins the tuned choices of this shape first
        buf = C.create_string_buffer(512)
        out = []
        for i in range(self._lib.rib_num_launches(self._h, B, H, W)):
            _native.check(self._h, self._lib.rib_debug_launch_info(self._h, B, H, W, i, buf, 512))
            name, kclass, grid, tile, flops, nbytes = buf.value.decode().split("|")
            out.append({"name": name, "class": int(kclass), "grid": grid, "tile": tile, "flops": float(flops), "bytes": float(nbytes)})
        return out
