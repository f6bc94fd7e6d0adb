"""ctypes binding of include/rib_motion.h (libribmotion.so).  No fallback: a missing library raises."""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(os.path.dirname(_HERE), "csrc", "libribmotion.so")


class RibmConfig(C.Structure):
    _fields_ = [(n, C.c_int32) for n in (
        "input_joints", "hidden_dim", "nheads", "dim_feedforward", "enc_layers", "dec_layers",
        "activation", "pre_norm", "two_stage")]


ACT_IDS = {"relu": 0, "gelu": 1, "leaky_relu": 2}


class RibmError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("ribm error %d: %s" % (code, msg))
        self.code = code


# name -> (restype, argtypes); must cover every symbol of include/rib_motion.h
SIGNATURES = {
    "ribm_create": (C.c_int, [C.POINTER(RibmConfig), C.c_int, C.POINTER(C.c_void_p)]),
    "ribm_destroy": (None, [C.c_void_p]),
    "ribm_last_error": (C.c_char_p, [C.c_void_p]),
    "ribm_build_info": (C.c_char_p, []),
    "ribm_num_tensors": (C.c_int, [C.c_void_p]),
    "ribm_tensor_info": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_char_p), C.POINTER(C.c_int), C.POINTER(C.c_int64)]),
    "ribm_set_tensor": (C.c_int, [C.c_void_p, C.c_char_p, C.c_void_p, C.c_int, C.POINTER(C.c_int64)]),
    "ribm_finalize_weights": (C.c_int, [C.c_void_p]),
    "ribm_weights_bytes": (C.c_size_t, [C.c_void_p]),
    "ribm_workspace_bytes": (C.c_size_t, [C.c_void_p, C.c_int, C.c_int]),
    "ribm_num_launches": (C.c_int, [C.c_void_p]),
    "ribm_forward": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int] + [C.c_void_p] * 9 + [C.c_size_t, C.c_void_p]),
}

_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError("%s is missing: build it with `python render-in-between_amd/csrc/build.py` "
                               "(there is no CPU fallback)" % LIB_PATH)
        l = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(l, name)
            fn.restype = res
            fn.argtypes = args
        _lib = l
    return _lib


def check(h, rc):
    if rc != 0:
        raise RibmError(rc, lib().ribm_last_error(h).decode())
