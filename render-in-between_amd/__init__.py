"""MI355X-native pose-guided neural-rendering generator (the hot path of
azuxmioy/Render-In-Between's Pose_Guided_Neural_Rendering), behind the
reference's own Generator / inference.py surface.

Host side: Python on PyTorch-ROCm (device memory, streams, torch.distributed).
Compute: hand-written HIP kernels for gfx950 in ``csrc/`` reached through the
C-ABI declared in ``include/rib.h`` (``librib.so``).  There is no CPU
fallback: using the generator without the built extension and a GPU raises.
"""
from .config import AttrDict, GenSpec, get_config, hsm_gen_config  # noqa: F401
from .spec import conv_inventory, state_dict_spec, conv_flops  # noqa: F401

__all__ = ["AttrDict", "GenSpec", "get_config", "hsm_gen_config",
           "conv_inventory", "state_dict_spec", "conv_flops", "Generator"]


def __getattr__(name):
    # Lazy: importing the package must not need the native library
    # (CPU-side tooling such as synth/spec is used by the oracle tests).
    if name == "Generator":
        from .generator import Generator
        return Generator
    raise AttributeError(name)
