"""Driver-boundary image helpers: the few third-party calls the reference drivers make either side of the hot path.

The reference calls scikit-image for these (absent here, and not part of the accelerated path); each function
below restates the published semantics of the call it replaces and cites the call site:

* ``im2double``          reference toolbox/imtools.py:42-53
* ``resize``             ``skimage.transform.resize(I, (h, w))`` as used at UnMicst1-5.py:815,850 (defaults: order 1,
                         mode='reflect', anti_aliasing when shrinking, preserve_range=False)
* ``rescale_intensity``  ``skimage.exposure.rescale_intensity(I, in_range=(lo, hi), out_range=(0, 0.983))``
                         (UnMicst1-5.py:821)

Pinning: at ``--scalingFactor 1`` (the only configuration with a golden output, "UNet sample data") resize is the
identity grid and the pipeline is pinned end to end by tests/test_oracle_golden.py / test_gpu_parity.py.  For other
scaling factors the interpolation follows scikit-image >= 0.19 (gaussian pre-filter + ``scipy.ndimage.zoom`` with
``grid_mode=True``); scikit-image is not installed here, so that branch is **unpinned**.
"""
from __future__ import annotations

import numpy as np


def im2double(I: np.ndarray) -> np.ndarray:
    if I.dtype == np.uint16:
        return I.astype(np.float64) / 65535
    if I.dtype == np.uint8:
        return I.astype(np.float64) / 255
    if I.dtype == np.float32:
        return I.astype(np.float64)
    return I


def img_as_float(I: np.ndarray) -> np.ndarray:
    """skimage's dtype conversion inside resize(preserve_range=False): unsigned ints are multiplied by 1/max."""
    if I.dtype.kind == "f":
        return I.astype(np.float64) if I.dtype == np.float16 else I
    if I.dtype.kind == "u":
        return np.multiply(I, 1.0 / np.iinfo(I.dtype).max, dtype=np.float64)
    if I.dtype.kind == "i":
        info = np.iinfo(I.dtype)
        out = np.multiply(I.astype(np.float64) - info.min, 2.0 / (float(info.max) - info.min))
        return out - 1.0
    if I.dtype == bool:
        return I.astype(np.float64)
    raise TypeError("unsupported image dtype %s" % I.dtype)


def resize(I: np.ndarray, output_shape) -> np.ndarray:
    """Bilinear resize with skimage.transform.resize's defaults; returns float in [0,1] for integer input."""
    out_shape = (int(output_shape[0]), int(output_shape[1]))
    if I.ndim != 2:
        raise ValueError("resize expects a 2-D plane")
    img = img_as_float(I)
    if tuple(img.shape) == out_shape:
        return np.array(img, copy=True)   # identity sampling grid: every output pixel reads exactly one input pixel
    from scipy import ndimage as ndi
    factors = np.divide(img.shape, out_shape)
    if any(o < i for o, i in zip(out_shape, img.shape)):
        sigma = np.maximum(0, (factors - 1) / 2)
        img = ndi.gaussian_filter(img, sigma, cval=0, mode="mirror")
    out = ndi.zoom(img, [1 / f for f in factors], order=1, mode="mirror", cval=0, grid_mode=True)
    if out.shape != out_shape:  # zoom rounds the shape itself; skimage passes the exact shape through `output`
        fixed = np.empty(out_shape, out.dtype)
        ndi.zoom(img, [1 / f for f in factors], output=fixed, order=1, mode="mirror", cval=0, grid_mode=True)
        out = fixed
    lo, hi = img.min(), img.max()
    return np.clip(out, lo, hi)


def rescale_intensity(I: np.ndarray, in_range, out_range) -> np.ndarray:
    imin, imax = float(in_range[0]), float(in_range[1])
    omin, omax = float(out_range[0]), float(out_range[1])
    I = np.clip(I, imin, imax)
    if imin != imax:
        I = (I - imin) / (imax - imin)
        return I * (omax - omin) + omin
    return np.clip(I, omin, omax)


def to_uint8_via_resize(pm_half: np.ndarray, raw_shape) -> np.ndarray:
    """The reference's output recipe (UnMicst1-5.py:848-854): uint8(255*pm) -> resize to the raw size (float =
    u8 * (1/255)) -> uint8(255 * .) -- truncation both times."""
    PM = np.uint8(255 * pm_half)          # float16 * int -> float16 product, truncated (numpy semantics kept)
    PM = resize(PM, raw_shape)
    return np.uint8(255 * PM)
