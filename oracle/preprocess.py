"""Oracle for the input side of the path.  TEST INFRASTRUCTURE ONLY (imported by tests/ alone).

`resize_linear_u8` restates, in numpy, the published algorithm of cv2.resize(src, (w, h), interpolation=INTER_LINEAR) on
8-bit images -- what albumentations' A.Resize(h, w) of the reference's val/test transform calls
(/root/reference/src/scripts/benchmark/models/skinLesionDatasets.py:116-120): half-pixel centres, float coefficients
rounded to 11-bit fixed point (INTER_RESIZE_COEF_BITS = 11), horizontal pass in 32-bit ints with the fraction zeroed at the
left / right border, vertical pass (((b0 * (S0 >> 4)) >> 16) + ((b1 * (S1 >> 4)) >> 16) + 2) >> 2 over rows clipped to the
image.  PARITY UNPINNED against cv2 itself: opencv / albumentations are not installed in the build container, so
the tests pin the HIP kernel to this restatement bit for bit and to plain float bilinear interpolation within one grey level.
The metadata encoding needs no restatement: sklearn (the reference's own dependency) is installed and is the oracle.
"""
import numpy as np


def _coef(dsize, ssize, horizontal):
    d = np.arange(dsize, dtype=np.float64)
    f = ((d + 0.5) * (ssize / dsize) - 0.5).astype(np.float32)
    s = np.floor(f).astype(np.int64)
    f = f - s.astype(np.float32)
    if horizontal:
        lo = s < 0
        f[lo] = 0.0; s[lo] = 0
        hi = s >= ssize - 1
        f[hi] = 0.0; s[hi] = ssize - 1
    c0 = np.rint((np.float32(1.0) - f) * np.float32(2048.0)).astype(np.int64)
    c1 = np.rint(f * np.float32(2048.0)).astype(np.int64)
    return np.clip(s, 0, ssize - 1), np.clip(s + 1, 0, ssize - 1), c0, c1


def resize_linear_u8(img, h, w):
    """img uint8 [..., Hs, Ws, C] -> uint8 [..., h, w, C]."""
    img = np.asarray(img)
    hs, ws = img.shape[-3], img.shape[-2]
    if (hs, ws) == (h, w):
        return img.copy()
    y0, y1, b0, b1 = _coef(h, hs, False)
    x0, x1, a0, a1 = _coef(w, ws, True)
    src = img.astype(np.int64)
    a0, a1 = a0[:, None], a1[:, None]
    rows = src[..., x0, :] * a0 + src[..., x1, :] * a1                   # horizontal pass: [..., Hs, w, C]
    s0, s1 = rows[..., y0, :, :], rows[..., y1, :, :]
    b0, b1 = b0[:, None, None], b1[:, None, None]
    v = (((b0 * (s0 >> 4)) >> 16) + ((b1 * (s1 >> 4)) >> 16) + 2) >> 2
    return np.clip(v, 0, 255).astype(np.uint8)


def resize_bilinear_float(img, h, w):
    """Plain float bilinear interpolation with half-pixel centres and replicated borders (cross-check, not bit-exact)."""
    img = np.asarray(img, dtype=np.float64)
    hs, ws = img.shape[-3], img.shape[-2]
    fy = np.clip((np.arange(h) + 0.5) * hs / h - 0.5, 0, hs - 1)
    fx = np.clip((np.arange(w) + 0.5) * ws / w - 0.5, 0, ws - 1)
    y0 = np.floor(fy).astype(int); x0 = np.floor(fx).astype(int)
    y1 = np.minimum(y0 + 1, hs - 1); x1 = np.minimum(x0 + 1, ws - 1)
    wy = (fy - y0)[:, None, None]; wx = (fx - x0)[None, :, None]
    top = img[..., y0, :, :][..., :, x0, :] * (1 - wx) + img[..., y0, :, :][..., :, x1, :] * wx
    bot = img[..., y1, :, :][..., :, x0, :] * (1 - wx) + img[..., y1, :, :][..., :, x1, :] * wx
    return top * (1 - wy) + bot * wy
