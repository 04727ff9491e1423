"""ctypes binding of the training step of libumx (include/umx_train.h) and the host-side mirror of the reference's
training regimes.

``Trainer.step(batchData, batchLabels, batchWeights)`` is one ``sess.run([optOp, loss], feed_dict=...)`` of the
reference's loop (UnMicst1-5.py:483-484, UnMicst2.py:471-472); ``solo_options`` / ``duo_options`` carry the constants
those scripts hard-code (optimiser, learning-rate schedule, regularisers, dropout rates, the probability clip).  The
trained parameters come back in the blob layout ``umx.Engine`` loads (``Trainer.blob()``), so train -> infer needs no
conversion.  No CPU fallback: without libumx and a gfx950 device every call raises.
"""
from __future__ import annotations

import ctypes
from dataclasses import dataclass, fields
from typing import Optional

import numpy as np

from . import umx as _umx
from .model import GRAPH_V2, HParams

OPT_ADAM, OPT_MOMENTUM = 0, 1
REG_NONE, REG_L1, REG_L2 = 0, 1, 2
TV_PARAMS, TV_GRADS, TV_SLOT_M, TV_SLOT_V = 0, 1, 2, 3

# every symbol include/umx_train.h declares (checked by tests/test_abi.py)
EXPORTS = [
    "umx_train_options_solo", "umx_train_options_duo", "umx_trainer_create", "umx_trainer_destroy",
    "umx_trainer_last_error", "umx_train_step", "umx_train_step_dev", "umx_trainer_loss", "umx_trainer_read",
    "umx_trainer_probs", "umx_trainer_read_tensor", "umx_trainer_eval", "umx_trainer_step_count", "umx_trainer_batch", "umx_trainer_flops_per_image",
    "umx_trainer_profile",
]


class _TrainOptions(ctypes.Structure):
    _fields_ = [("device_ordinal", ctypes.c_int32), ("batch", ctypes.c_int32), ("optimizer", ctypes.c_int32),
                ("decay_steps", ctypes.c_int32), ("lr0", ctypes.c_float), ("decay_rate", ctypes.c_float),
                ("momentum", ctypes.c_float), ("beta1", ctypes.c_float), ("beta2", ctypes.c_float),
                ("adam_eps", ctypes.c_float), ("reg_kind", ctypes.c_int32), ("reg_down", ctypes.c_float),
                ("reg_bottom", ctypes.c_float), ("reg_up", ctypes.c_float), ("reg_top", ctypes.c_float),
                ("clip_eps", ctypes.c_float), ("drop_down_step", ctypes.c_float), ("drop_bottom", ctypes.c_float),
                ("drop_up0", ctypes.c_float), ("drop_up_step", ctypes.c_float), ("bn_momentum", ctypes.c_float),
                ("seed", ctypes.c_uint64), ("reserved", ctypes.c_int32 * 8)]


@dataclass
class TrainOptions:
    """Field-for-field ``umx_train_options``; the defaults are the solo script's (UnMicst1-5.py:84,139,362-378)."""
    optimizer: int = OPT_ADAM
    lr0: float = 5e-5
    decay_steps: int = 5000
    decay_rate: float = 0.98
    momentum: float = 0.9
    beta1: float = 0.9
    beta2: float = 0.999
    adam_eps: float = 1e-8
    reg_kind: int = REG_L1
    reg_down: float = 8e-5
    reg_bottom: float = 8e-5
    reg_up: float = 8e-5
    reg_top: float = 8e-5
    clip_eps: float = 1e-7
    drop_down_step: float = 0.0
    drop_bottom: float = 0.35
    drop_up0: float = 0.0
    drop_up_step: float = 0.0
    bn_momentum: float = 0.99
    seed: int = 1234


def solo_options(**kw) -> TrainOptions:
    return TrainOptions(**kw)


def duo_options(**kw) -> TrainOptions:
    """UnMicst2.py:82,114,123,137,158,203,211,357-371."""
    base = dict(lr0=6e-5, decay_steps=4000, decay_rate=0.99, reg_kind=REG_L2, reg_down=0.01, reg_bottom=0.01,
                reg_up=0.005, reg_top=0.005, clip_eps=0.0, drop_down_step=0.05, drop_bottom=0.3, drop_up0=0.25,
                drop_up_step=0.05)
    base.update(kw)
    return TrainOptions(**base)


def _bind(L):
    if getattr(L, "_umx_train_bound", False):
        return L
    c_int, c_void_p = ctypes.c_int, ctypes.c_void_p
    dp = ctypes.POINTER(ctypes.c_double)
    L.umx_train_options_solo.restype = None
    L.umx_train_options_solo.argtypes = [ctypes.POINTER(_TrainOptions)]
    L.umx_train_options_duo.restype = None
    L.umx_train_options_duo.argtypes = [ctypes.POINTER(_TrainOptions)]
    L.umx_trainer_create.restype = c_int
    L.umx_trainer_create.argtypes = [ctypes.POINTER(_umx._HP), c_void_p, ctypes.c_size_t, ctypes.POINTER(_TrainOptions),
                                     ctypes.POINTER(c_void_p)]
    L.umx_trainer_destroy.restype = None
    L.umx_trainer_destroy.argtypes = [c_void_p]
    L.umx_trainer_last_error.restype = ctypes.c_char_p
    L.umx_trainer_last_error.argtypes = [c_void_p]
    L.umx_train_step.restype = c_int
    L.umx_train_step.argtypes = [c_void_p, c_void_p, c_void_p, c_void_p, c_int, dp]
    L.umx_train_step_dev.restype = c_int
    L.umx_train_step_dev.argtypes = [c_void_p, c_void_p, c_void_p, c_void_p, c_int]
    L.umx_trainer_loss.restype = c_int
    L.umx_trainer_loss.argtypes = [c_void_p, dp]
    L.umx_trainer_read.restype = c_int
    L.umx_trainer_read.argtypes = [c_void_p, c_int, c_void_p, ctypes.c_size_t]
    L.umx_trainer_probs.restype = c_int
    L.umx_trainer_probs.argtypes = [c_void_p, c_void_p]
    L.umx_trainer_read_tensor.restype = c_int
    L.umx_trainer_read_tensor.argtypes = [c_void_p, ctypes.c_char_p, c_void_p, ctypes.POINTER(ctypes.c_size_t)]
    L.umx_trainer_eval.restype = c_int
    L.umx_trainer_eval.argtypes = [c_void_p, c_void_p, c_void_p]
    L.umx_trainer_step_count.restype = ctypes.c_int64
    L.umx_trainer_step_count.argtypes = [c_void_p]
    L.umx_trainer_batch.restype = c_int
    L.umx_trainer_batch.argtypes = [c_void_p]
    L.umx_trainer_flops_per_image.restype = ctypes.c_double
    L.umx_trainer_flops_per_image.argtypes = [c_void_p]
    L.umx_trainer_profile.restype = c_int
    L.umx_trainer_profile.argtypes = [c_void_p, c_int, dp, dp, dp, ctypes.POINTER(c_int)]
    L._umx_train_bound = True
    return L


def native_options(kind: str) -> TrainOptions:
    """The presets as libumx fills them (umx_train_options_solo/_duo) -- tests compare them with the dataclasses."""
    L = _bind(_umx.load())
    o = _TrainOptions()
    (L.umx_train_options_solo if kind == "solo" else L.umx_train_options_duo)(ctypes.byref(o))
    return TrainOptions(**{f.name: getattr(o, f.name) for f in fields(TrainOptions)})


class Trainer:
    def __init__(self, hp: HParams, blob: np.ndarray, opts: Optional[TrainOptions] = None, batch: int = 0,
                 device: int = 0):
        if hp.graph != GRAPH_V2 or hp.nExtraConvs != 0:
            raise ValueError("the training step covers the v2 graph with nExtraConvs == 0")
        self._lib = _bind(_umx.load())
        self.hp = hp
        self.opts = opts or TrainOptions()
        o = _TrainOptions()
        for f in fields(TrainOptions):
            setattr(o, f.name, getattr(self.opts, f.name))
        o.device_ordinal = int(device)
        o.batch = int(batch) if batch else int(hp.batchSize)
        blob = np.ascontiguousarray(blob, dtype=np.float32)
        self.nparams = int(blob.size)
        h = ctypes.c_void_p()
        hps = _umx._hp_struct(hp)
        rc = self._lib.umx_trainer_create(ctypes.byref(hps), blob.ctypes.data, blob.size, ctypes.byref(o), ctypes.byref(h))
        if rc != _umx.UMX_OK:
            raise _umx.UmxError(rc, (self._lib.umx_trainer_last_error(None) or b"").decode())
        self._h = h
        self.batch = int(self._lib.umx_trainer_batch(h))
        self.device = int(device)

    # ------------------------------------------------------------------------------------------
    def _check(self, rc):
        if rc != _umx.UMX_OK:
            raise _umx.UmxError(rc, (self._lib.umx_trainer_last_error(self._h) or b"").decode())

    def close(self):
        if getattr(self, "_h", None):
            self._lib.umx_trainer_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _batch_arrays(self, data, labels, weights):
        hp, B = self.hp, self.batch
        d = np.ascontiguousarray(data, dtype=np.float32)
        y = np.ascontiguousarray(labels, dtype=np.float32)
        w = np.ascontiguousarray(weights, dtype=np.float32)
        if d.shape != (B, hp.imSize, hp.imSize, hp.nChannels):
            raise ValueError("data must be %r, got %r" % ((B, hp.imSize, hp.imSize, hp.nChannels), d.shape))
        if y.shape != (B, hp.imSize, hp.imSize, hp.nClasses) or w.shape != y.shape:
            raise ValueError("labels / weights must be %r" % ((B, hp.imSize, hp.imSize, hp.nClasses),))
        return d, y, w

    def step(self, data, labels, weights, apply_update: bool = True):
        """-> (loss, data term, regularisation loss) of this batch; parameters updated unless apply_update is False."""
        d, y, w = self._batch_arrays(data, labels, weights)
        out = (ctypes.c_double * 3)()
        self._check(self._lib.umx_train_step(self._h, d.ctypes.data, y.ctypes.data, w.ctypes.data, int(apply_update), out))
        return float(out[0]), float(out[1]), float(out[2])

    def step_dev(self, data_ptr: int, labels_ptr: int, weights_ptr: int, apply_update: bool = True):
        """Device pointers (e.g. torch tensors' data_ptr()); only enqueues -- call loss() to synchronise."""
        self._check(self._lib.umx_train_step_dev(self._h, ctypes.c_void_p(data_ptr), ctypes.c_void_p(labels_ptr),
                                                 ctypes.c_void_p(weights_ptr), int(apply_update)))

    def loss(self):
        out = (ctypes.c_double * 3)()
        self._check(self._lib.umx_trainer_loss(self._h, out))
        return float(out[0]), float(out[1]), float(out[2])

    def _read(self, which: int) -> np.ndarray:
        out = np.empty(self.nparams, np.float32)
        self._check(self._lib.umx_trainer_read(self._h, which, out.ctypes.data, out.size))
        return out

    def blob(self) -> np.ndarray:
        """Current variables (incl. BN moving statistics) in the weight-blob layout of umx.Engine."""
        return self._read(TV_PARAMS)

    def grads(self) -> np.ndarray:
        return self._read(TV_GRADS)

    def slots(self):
        return self._read(TV_SLOT_M), self._read(TV_SLOT_V)

    def read_tensor(self, name: str) -> np.ndarray:
        """A tensor of the last step's forward pass as a flat float32 array (umx_trainer_read_tensor: "ld0.z", "ld0.stat", "lu1.us",
        "ds2", ...; NHWC order)."""
        n = ctypes.c_size_t(0)
        self._check(self._lib.umx_trainer_read_tensor(self._h, name.encode(), None, ctypes.byref(n)))
        out = np.empty(n.value, dtype=np.float32)
        self._check(self._lib.umx_trainer_read_tensor(self._h, name.encode(), out.ctypes.data, ctypes.byref(n)))
        return out

    def probs(self) -> np.ndarray:
        hp = self.hp
        out = np.empty((self.batch, hp.imSize, hp.imSize, hp.nClasses), np.float32)
        self._check(self._lib.umx_trainer_probs(self._h, out.ctypes.data))
        return out

    def eval(self, data) -> np.ndarray:
        """Inference-mode forward of one batch with the current variables (``tfTraining: 0``) -> probabilities."""
        hp = self.hp
        d = np.ascontiguousarray(data, dtype=np.float32)
        if d.shape != (self.batch, hp.imSize, hp.imSize, hp.nChannels):
            raise ValueError("data must be %r, got %r" % ((self.batch, hp.imSize, hp.imSize, hp.nChannels), d.shape))
        out = np.empty((self.batch, hp.imSize, hp.imSize, hp.nClasses), np.float32)
        self._check(self._lib.umx_trainer_eval(self._h, d.ctypes.data, out.ctypes.data))
        return out

    @property
    def step_count(self) -> int:
        return int(self._lib.umx_trainer_step_count(self._h))

    @property
    def flops_per_image(self) -> float:
        return float(self._lib.umx_trainer_flops_per_image(self._h))

    def profile(self, enable: bool = True) -> dict:
        f, b, o = ctypes.c_double(), ctypes.c_double(), ctypes.c_double()
        n = ctypes.c_int()
        self._check(self._lib.umx_trainer_profile(self._h, int(enable), ctypes.byref(f), ctypes.byref(b), ctypes.byref(o),
                                                  ctypes.byref(n)))
        return {"forward_ms": f.value, "backward_ms": b.value, "update_ms": o.value, "steps": n.value}
