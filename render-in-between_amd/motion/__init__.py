"""Stage 1 of the pipeline on the MI355X: the motion transformer that turns
low-frame-rate OpenPose key frames into the interpolated pose sequence (``Predict_motion/``) the
generator is conditioned on.  ``HMM`` = the reference's Human_Motion_Modelling directory.

Host side mirrors the reference's inference surface (``Model_inference.inference``, the
``--config / --save-dir / --pose-dir / --upsample-rate`` CLI, the ``Predict_motion`` /
``Linear_motion`` output folders); the network runs as hand-written HIP kernels behind the C ABI of
``include/rib_motion.h`` (``libribmotion.so``).  No CPU fallback.
"""
from .spec import MotionSpec, state_dict_spec  # noqa: F401

__all__ = ["MotionSpec", "state_dict_spec", "MotionTransformer", "ModelInference"]


def __getattr__(name):
    if name in ("MotionTransformer", "ModelInference", "PositionEmbeddingSine1D"):
        from . import model
        return getattr(model, name)
    raise AttributeError(name)
