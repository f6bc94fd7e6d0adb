"""Frame resize of the folder driver: OpenCV's 8-bit INTER_CUBIC, restated.

The reference resizes every key frame / DAIN frame to the model size with `albumentations.Resize(interpolation=
cv2.INTER_CUBIC)` (PGNR/models/evaluator.py:18-26, applied at :219-221), i.e. `cv2.resize(img, (W, H),
interpolation=cv2.INTER_CUBIC)` on a uint8 HWC array.  cv2 / albumentations are third-party dependencies that are not
in this image (opencv-python, unpinned in the reference's requirements), so the algorithm is restated here from
OpenCV 4.x's published implementation (modules/imgproc/src/resize.cpp: `resizeGeneric_` with `HResizeCubic`,
`VResizeCubic`, `FixedPtCast<int, uchar, INTER_RESIZE_COEF_BITS * 2>`, `interpolateCubic` with A = -0.75):

* pixel centres: source coordinate fx = (dx + 0.5) * (src / dst) - 0.5 (computed in double, kept as float32),
  sx = floor(fx), the four taps sx - 1 .. sx + 2 with indices clamped to the image (replicated border);
* Keys cubic weights with A = -0.75 in float32, the fourth as 1 - (w0 + w1 + w2), each converted to a 16-bit fixed-point
  coefficient round-to-nearest-even(w * 2048);
* horizontal pass in int32 without rounding, vertical pass in int32, then (v + 2^21) >> 22 saturated to uint8;
* NO low-pass filter when the image shrinks (PIL's `BICUBIC` widens its kernel on reduction, and uses A = -0.5: a
  1080p frame reduced to 512x512 differs visibly between the two).

**Parity unpinned**: there is no cv2 here to compare with and the reference holds no fixture for this step.  Known
residual: OpenCV's SIMD builds evaluate the vertical pass in float32 (`VResizeCubicVec_32s8u`) and can differ from the
fixed-point definition restated here by one grey level on isolated pixels.  `oracle/resize_ref.py` is the slow scalar
statement of the same definition (written independently, pixel by pixel) that the tests hold this version to.
"""
import functools

import numpy as np

_COEF_BITS = 11
_COEF_SCALE = 1 << _COEF_BITS


def _cubic_taps(dst, src):
    """Tap indices [dst, 4] (clamped) and fixed-point weights [dst, 4] (int32 holding shorts) along one axis."""
    scale = np.float64(1.0) / (np.float64(dst) / np.float64(src))          # resize.cpp: scale_x = 1. / inv_scale_x
    d = np.arange(dst, dtype=np.float64)
    f = ((d + 0.5) * scale - 0.5).astype(np.float32)
    s = np.floor(f).astype(np.int64)
    x = (f - s.astype(np.float32)).astype(np.float32)
    A = np.float32(-0.75)
    one = np.float32(1.0)
    w0 = ((A * (x + one) - np.float32(5.0) * A) * (x + one) + np.float32(8.0) * A) * (x + one) - np.float32(4.0) * A
    w1 = ((A + np.float32(2.0)) * x - (A + np.float32(3.0))) * x * x + one
    xr = one - x
    w2 = ((A + np.float32(2.0)) * xr - (A + np.float32(3.0))) * xr * xr + one
    w3 = one - w0 - w1 - w2
    w = np.stack([w0, w1, w2, w3], axis=1).astype(np.float32)
    coef = np.rint(w * np.float32(_COEF_SCALE)).astype(np.int32)          # saturate_cast<short>(float): nearest-even
    coef = np.clip(coef, -32768, 32767)
    idx = np.clip(s[:, None] + np.arange(-1, 3)[None, :], 0, src - 1)
    return idx, coef


@functools.lru_cache(maxsize=32)
def _tap_matrix(dst, src):
    """The taps of one axis as a CSR matrix [dst, src] (clamped taps that coincide add up, as in HResizeCubic).
    Cached per (dst, src): a clip resizes every frame between the same two sizes, on several decode threads (the
    matrix is only read afterwards).  scipy is a dependency of the folder driver (declared in README)."""
    from scipy import sparse
    idx, coef = _cubic_taps(dst, src)
    rows = np.repeat(np.arange(dst), 4)
    return sparse.csr_matrix((coef.reshape(-1).astype(np.int32), (rows, idx.reshape(-1))), shape=(dst, src))


def resize_cubic_u8(img, width, height):
    """uint8 [H0, W0] or [H0, W0, C] -> uint8 [height, width(, C)], OpenCV INTER_CUBIC semantics (module docstring)."""
    a = np.asarray(img)
    if a.dtype != np.uint8:
        raise TypeError("resize_cubic_u8 expects a uint8 image")
    squeeze = a.ndim == 2
    if squeeze:
        a = a[:, :, None]
    h0, w0, _ = a.shape
    if (h0, w0) == (height, width):
        out = a.copy()
        return out[:, :, 0] if squeeze else out
    # both passes are products with a 4-per-row sparse integer matrix, exact in int32 exactly as in OpenCV
    # (|v| <= 255 * (1.375 * 2048)^2 < 2^31), so their order is free: the pass that shrinks the rows runs first on the
    # image's own [H0][W0*C] layout and the transposes only see the reduced array (1080p -> 512x512: 20 ms)
    c = a.shape[2]
    wy = _tap_matrix(height, h0)                                           # [H, H0]
    wx = _tap_matrix(width, w0)                                            # [W, W0]
    v1 = wy.dot(a.reshape(h0, w0 * c).astype(np.int32))                    # [H, W0*C]
    cols = np.ascontiguousarray(v1.reshape(height, w0, c).transpose(1, 0, 2)).reshape(w0, height * c)
    v2 = wx.dot(cols).reshape(width, height, c)                            # [W, H, C]
    v = np.ascontiguousarray(v2.transpose(1, 0, 2)).astype(np.int64)
    v = (v + (1 << (2 * _COEF_BITS - 1))) >> (2 * _COEF_BITS)             # FixedPtCast<int, uchar, 22>
    out = np.clip(v, 0, 255).astype(np.uint8)
    return out[:, :, 0] if squeeze else out
