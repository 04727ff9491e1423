"""ORACLE -- TEST INFRASTRUCTURE ONLY (never imported by unmicst_amd/).

ctypes binding of oracle/unet_oracle.c (CPU restatement of the reference's UNet forward) plus the glue that
makes a full reference-equivalent ``singleImageInference``.  Importers: tests/, __graft_entry__.smoke(),
bench.py's cpu_baseline leg -- as the checker / CPU baseline only.
"""
from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np

from . import pi2d_oracle

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "_build", "libunet_oracle.so")
_lib = None


class _HP(ctypes.Structure):
    _fields_ = [(n, ctypes.c_int) for n in
                ("graph", "imSize", "nChannels", "nClasses", "nOut0", "nLayers", "ks", "nExtraConvs", "featMapsFact")]


def build(force: bool = False) -> str:
    """Compile the C restatement (gcc + OpenMP) into oracle/_build/."""
    src = os.path.join(_HERE, "unet_oracle.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s"] + (["-B"] if force else []))
    return _LIB_PATH


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        L = ctypes.CDLL(_LIB_PATH)
        fp = ctypes.POINTER(ctypes.c_float)
        L.orc_forward.restype = ctypes.c_int
        L.orc_forward.argtypes = [ctypes.POINTER(_HP), fp, ctypes.c_size_t, fp, ctypes.c_int, fp]
        L.orc_conv2d_same.restype = None
        L.orc_conv2d_same.argtypes = [fp] + [ctypes.c_int] * 4 + [fp] + [ctypes.c_int] * 3 + [fp]
        L.orc_conv2d_transpose_s2.restype = None
        L.orc_conv2d_transpose_s2.argtypes = [fp] + [ctypes.c_int] * 4 + [fp] + [ctypes.c_int] * 3 + [fp]
        L.orc_num_threads.restype = ctypes.c_int
        L.orc_set_num_threads.argtypes = [ctypes.c_int]
        _lib = L
    return _lib


def _fp(a: np.ndarray):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_float))


def _hp_struct(hp) -> _HP:
    return _HP(int(hp.graph), int(hp.imSize), int(hp.nChannels), int(hp.nClasses), int(hp.nOut0),
               int(hp.nLayers), int(hp.ks), int(hp.nExtraConvs), int(hp.featMapsFact))


def num_threads() -> int:
    return int(lib().orc_num_threads())


def set_num_threads(n: int) -> None:
    lib().orc_set_num_threads(int(n))


def forward(hp, blob: np.ndarray, x: np.ndarray) -> np.ndarray:
    """UNet forward on a normalised float32 NHWC batch -> softmax probabilities [B,P,P,K] (== Session.run)."""
    x = np.ascontiguousarray(x, dtype=np.float32)
    blob = np.ascontiguousarray(blob, dtype=np.float32)
    B = x.shape[0]
    assert x.shape[1:] == (hp.imSize, hp.imSize, hp.nChannels), x.shape
    out = np.empty((B, hp.imSize, hp.imSize, hp.nClasses), np.float32)
    h = _hp_struct(hp)
    rc = lib().orc_forward(ctypes.byref(h), _fp(blob), blob.size, _fp(x), B, _fp(out))
    if rc != 0:
        raise RuntimeError("orc_forward failed with code %d" % rc)
    return out


def conv2d_same(x: np.ndarray, w: np.ndarray) -> np.ndarray:
    x = np.ascontiguousarray(x, np.float32)
    w = np.ascontiguousarray(w, np.float32)
    B, H, W, Cin = x.shape
    kh, kw, ci, Cout = w.shape
    assert ci == Cin
    y = np.empty((B, H, W, Cout), np.float32)
    lib().orc_conv2d_same(_fp(x), B, H, W, Cin, _fp(w), kh, kw, Cout, _fp(y))
    return y


def conv2d_transpose_s2(x: np.ndarray, wt: np.ndarray) -> np.ndarray:
    x = np.ascontiguousarray(x, np.float32)
    wt = np.ascontiguousarray(wt, np.float32)
    B, h, w, Cin = x.shape
    kh, kw, Cout, ci = wt.shape
    assert ci == Cin
    y = np.empty((B, 2 * h, 2 * w, Cout), np.float32)
    lib().orc_conv2d_transpose_s2(_fp(x), B, h, w, Cin, _fp(wt), kh, kw, Cout, _fp(y))
    return y


def tile_probs(hp, blob, image: np.ndarray, mean: float, std: float, duplicate_plane: bool = False,
               batch_size: int = 16) -> np.ndarray:
    """Per-tile softmax outputs [T,P,P,K] for every PI2D patch of ``image`` (row-major patch order)."""
    pi = pi2d_oracle.PI2DOracle(image, hp.imSize, int(hp.imSize / 8), "accumulate")
    outs = []
    i = 0
    while i < pi.num_patches:
        n = min(batch_size, pi.num_patches - i)
        outs.append(forward(hp, blob, pi2d_oracle.normalised_batch(pi, i, n, hp.nChannels, mean, std,
                                                                   duplicate_plane)))
        i += n
    return np.concatenate(outs)


def single_image_inference(hp, blob, image: np.ndarray, mean: float, std: float, mode: str, pm_index: int,
                           duplicate_plane: bool = False, batch_size: int = 16) -> np.ndarray:
    """Reference-equivalent UNet2D.singleImageInference (UnMicst1-5.py:687-710): fp16 plane of one class."""
    return pi2d_oracle.single_image_inference(
        image, lambda b: forward(hp, blob, b), hp.imSize, hp.nChannels, mean, std, mode, pm_index,
        batch_size, duplicate_plane)
