"""TEST INFRASTRUCTURE.  Import the *real* reference generator from
/root/reference (build container only — the reference never travels to the
GPU box).  Used solely by tests/golden/make_golden.py to pin the oracle and
to produce the committed fixtures.  Recipe: SURVEY.md Appendix A."""
import os
import sys
import types

REF_ROOT = "/root/reference/Pose_Guided_Neural_Rendering"


def available() -> bool:
    return os.path.isfile(os.path.join(REF_ROOT, "models", "generator.py"))


class _EasyDict(dict):
    """Minimal easydict.EasyDict: vars(obj) must expose the keys, because the
    reference splats sub-configs with ``**vars(params)`` (layers/conv.py:48-52)."""

    def __init__(self, d=None):
        super().__init__()
        for k, v in dict(d or {}).items():
            setattr(self, k, v)

    def __setattr__(self, k, v):
        if isinstance(v, dict) and not isinstance(v, _EasyDict):
            v = _EasyDict(v)
        super().__setattr__(k, v)
        super().__setitem__(k, v)

    __setitem__ = __setattr__


def _to_plain(d):
    return {k: (_to_plain(v) if isinstance(v, dict) else v) for k, v in dict(d).items()}


def load_reference_generator(gen_cfg: dict):
    """Build the reference ``Generator(gen_cfg)`` in eval mode."""
    if not available():
        raise RuntimeError("reference tree not present (only exists in the build container)")
    if "easydict" not in sys.modules:
        m = types.ModuleType("easydict")
        m.EasyDict = _EasyDict
        sys.modules["easydict"] = m            # utils/utils.py:7
    if "patoolib" not in sys.modules:
        sys.modules["patoolib"] = types.ModuleType("patoolib")   # utils/utils.py:6
    if REF_ROOT not in sys.path:
        sys.path.insert(0, REF_ROOT)
    from models.generator import Generator    # noqa: E402  (the reference's)
    return Generator(_EasyDict(_to_plain(gen_cfg))).eval()
