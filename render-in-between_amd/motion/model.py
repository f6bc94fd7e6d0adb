"""Host-side mirror of the reference's stage-1 module protocol
(HMM/models/transformer.py:Transformer, HMM/models/position_encoding.py:PositionEmbeddingSine_1D,
HMM/inference.py:Model_inference):

    transformer = MotionTransformer(cfg.transformer)       # build_transformer(cfg.transformer)
    transformer.load_state_dict(state_dict)                # strict, the reference's 188 tensors
    model = ModelInference(PositionEmbeddingSine1D(64), transformer)
    pred = model.inference(data, interp, encoder_mask, decoder_mask, rate)     # [1][C][L]

All compute runs in the HIP kernels behind include/rib_motion.h; this module validates arguments,
owns the workspace and passes device pointers.  No CPU fallback.
"""
from __future__ import annotations

import ctypes as C
import math
from typing import Dict, Optional

import torch

from . import _native
from .spec import MotionSpec


def _ptr(t: Optional[torch.Tensor]):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


class PositionEmbeddingSine1D:
    """PositionEmbeddingSine_1D (HMM/models/position_encoding.py:10-50), normalize=True.  The table
    depends only on (N, L): like the reference (which fills a host tensor, position_encoding.py:42-44)
    it is built on the host with torch ops and cached per shape on the device."""

    def __init__(self, num_pos_feats=64, temperature=10000, normalize=True, scale=None):
        if scale is not None and normalize is False:
            raise ValueError("normalize should be True if scale is passed")
        self.num_pos_feats = num_pos_feats
        self.temperature = temperature
        self.normalize = normalize
        self.scale = 2 * math.pi if scale is None else scale
        self._cache: Dict[tuple, torch.Tensor] = {}

    def eval(self):
        return self

    def to(self, *a, **k):
        return self

    def __call__(self, mask: torch.Tensor) -> torch.Tensor:
        N, L = mask.shape
        key = (N, L, str(mask.device))
        pe = self._cache.get(key)
        if pe is None:
            position = torch.arange(0, L, dtype=torch.float32).unsqueeze(0).repeat(N, 1)
            if self.normalize:
                position = position / (position[:, -1:] + 1e-6) * self.scale
            dim_t = torch.arange(self.num_pos_feats, dtype=torch.float32)
            dim_t = self.temperature ** (2 * (dim_t // 2) / self.num_pos_feats)
            pe = torch.zeros(N, L, self.num_pos_feats * 2)
            pe[:, :, 0::2] = torch.sin(position[:, :, None] / dim_t)
            pe[:, :, 1::2] = torch.cos(position[:, :, None] / dim_t)
            pe = pe.permute(1, 0, 2).contiguous().to(mask.device)
            self._cache[key] = pe
        return pe


class MotionTransformer:
    def __init__(self, cfg, device=None):
        self.spec = cfg if isinstance(cfg, MotionSpec) else MotionSpec.from_cfg(cfg)
        self.spec.validate()
        if not torch.cuda.is_available():
            raise RuntimeError("render_in_between_amd.motion.MotionTransformer needs a ROCm GPU (MI355X); there is no CPU path")
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("MotionTransformer device must be a GPU, got %s" % (self.device,))
        if self.device.index is None:
            self.device = torch.device("cuda", torch.cuda.current_device())
        self._lib = _native.lib()
        s = self.spec
        cfg_c = _native.RibmConfig(input_joints=s.input_joints, hidden_dim=s.hidden_dim, nheads=s.nheads,
                                   dim_feedforward=s.dim_feedforward, enc_layers=s.enc_layers, dec_layers=s.dec_layers,
                                   activation=_native.ACT_IDS[s.activation], pre_norm=int(s.pre_norm), two_stage=int(s.two_stage))
        h = C.c_void_p()
        rc = self._lib.ribm_create(C.byref(cfg_c), self.device.index, C.byref(h))
        if rc != 0:
            msg = self._lib.ribm_last_error(None).decode()
            raise (NotImplementedError("ribm_create: " + msg) if rc == -2 else _native.RibmError(rc, msg))
        self._h = h
        self._ws: Dict[tuple, torch.Tensor] = {}
        self.d_model = s.hidden_dim
        self.nhead = s.nheads
        self.two_stage = s.two_stage
        self.training = False

    def __del__(self):
        h = getattr(self, "_h", None)
        if h:
            self._lib.ribm_destroy(h)
            self._h = None

    def eval(self):
        self.training = False
        return self

    def train(self, mode=True):
        if mode:
            raise NotImplementedError("the MI355X path is inference-only")
        return self.eval()

    def to(self, *a, **k):
        return self

    def expected_tensors(self):
        out = []
        name = C.c_char_p(); ndim = C.c_int(); dims = (C.c_int64 * 2)()
        for i in range(self._lib.ribm_num_tensors(self._h)):
            _native.check(self._h, self._lib.ribm_tensor_info(self._h, i, C.byref(name), C.byref(ndim), dims))
            out.append((name.value.decode(), tuple(dims[j] for j in range(ndim.value))))
        return out

    def load_state_dict(self, state_dict, strict=True):
        """nn.Module.load_state_dict(strict=True) as reached from load_state_dict(net, path)
        (HMM/utils/utils.py:66-80): optional 'state_dict' wrapper and 'module.' prefixes removed."""
        if "state_dict" in state_dict and not torch.is_tensor(state_dict["state_dict"]):
            state_dict = state_dict["state_dict"]
        sd = {k.replace("module.", ""): v for k, v in state_dict.items()}
        expected = self.expected_tensors()
        names = {n for n, _ in expected}
        missing = sorted(names - set(sd)); unexpected = sorted(set(sd) - names)
        if strict and (missing or unexpected):
            raise RuntimeError("Error(s) in loading state_dict for Transformer:\n\tMissing key(s): %s\n\tUnexpected key(s): %s"
                               % (missing[:8], unexpected[:8]))
        for n, shape in expected:
            if n not in sd:
                continue
            t = sd[n].detach().to("cpu", torch.float32).contiguous()
            if tuple(t.shape) != shape:
                raise RuntimeError("size mismatch for %s: copying a param with shape %s, the model expects %s"
                                   % (n, tuple(t.shape), shape))
            d = (C.c_int64 * t.dim())(*t.shape)
            _native.check(self._h, self._lib.ribm_set_tensor(self._h, n.encode(), C.c_void_p(t.data_ptr()), t.dim(), d))
        with torch.cuda.device(self.device):
            _native.check(self._h, self._lib.ribm_finalize_weights(self._h))
        return self

    def _workspace(self, N, L):
        key = (N, L)
        ws = self._ws.get(key)
        if ws is None:
            n = int(self._lib.ribm_workspace_bytes(self._h, N, L))
            ws = torch.empty(n, dtype=torch.uint8, device=self.device)
            self._ws[key] = ws
        return ws

    def forward(self, src, src_mask, src_pos, tgt, tgt_mask, tgt_pos, rate):
        """Transformer.forward (HMM/models/transformer.py:78-111): src/tgt [N][C][L], masks bool [N][L],
        pos [L][N][D] -> (joints [L][N][C], reco [L][N][C]), all on this device."""
        def dev(t, dt):
            return t.to(self.device, dt).contiguous()
        src = dev(src, torch.float32)
        N, Cj, L = src.shape
        if Cj != self.spec.input_joints:
            raise RuntimeError("src has %d channels, the model expects %d" % (Cj, self.spec.input_joints))
        tgt = dev(tgt, torch.float32) if tgt is not None else None
        sm = dev(src_mask, torch.uint8); tm = dev(tgt_mask, torch.uint8)
        sp = dev(src_pos, torch.float32); tp = dev(tgt_pos, torch.float32)
        D = self.spec.hidden_dim
        for name, t, shape in (("src_mask", sm, (N, L)), ("tgt_mask", tm, (N, L)), ("src_pos", sp, (L, N, D)), ("tgt_pos", tp, (L, N, D))):
            if tuple(t.shape) != shape:
                raise RuntimeError("%s has shape %s, expected %s" % (name, tuple(t.shape), shape))
        if tgt is not None and tuple(tgt.shape) != (N, Cj, L):
            raise RuntimeError("tgt has shape %s, expected %s" % (tuple(tgt.shape), (N, Cj, L)))
        joints = torch.empty((L, N, Cj), dtype=torch.float32, device=self.device)
        reco = torch.empty((L, N, Cj), dtype=torch.float32, device=self.device)
        ws = self._workspace(N, L)
        with torch.cuda.device(self.device):
            st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
            _native.check(self._h, self._lib.ribm_forward(self._h, N, L, int(rate), _ptr(src), _ptr(sm), _ptr(sp), _ptr(tgt), _ptr(tm),
                                                          _ptr(tp), _ptr(joints), _ptr(reco), _ptr(ws), ws.numel(), st))
        return joints, reco

    __call__ = forward


class ModelInference:
    """Model_inference (HMM/inference.py:12-41)."""

    def __init__(self, enc, transformer):
        self.pos_encode = enc
        self.transformer = transformer
        self.device = transformer.device

    def inference(self, data, interp, encoder_mask, decoder_mask, rate):
        self.pos_encode.eval()
        self.transformer.eval()
        src = torch.unsqueeze(data, dim=0).to(self.device)
        tgt = torch.unsqueeze(interp, dim=0).to(self.device)
        src_mask = torch.unsqueeze(encoder_mask, dim=0).to(self.device)
        tgt_mask = torch.unsqueeze(decoder_mask, dim=0).to(self.device)
        pred, _ = self.transformer.forward(src, src_mask, self.pos_encode(src_mask), tgt, tgt_mask, self.pos_encode(tgt_mask), rate)
        return pred.permute(1, 2, 0)
