"""``UNet2D`` -- host-side mirror of the reference's namespace-class for the inference path.

The reference keeps the whole path behind three static methods of a class used as a namespace, with module-level
state (reference UnMicst1-5.py:33-40,656-710; same surface in UnMicst2.py:635-689 and UnMicst.py:489-541):

    UNet2D.singleImageInferenceSetup(modelPath, gpuIndex, mean, std)
    UNet2D.singleImageInference(image, mode, pmIndex)  -> float16 (H, W)
    UNet2D.singleImageInferenceCleanup()

This module keeps those names, argument meanings (``-1`` mean/std = use the model's pickled scalars, ``mode`` in
{'accumulate', 'replace'}, image float64 (H, W) or channel-first (C, H, W)) and the single-threaded, one-model-at-a-
time behaviour, and routes everything to libumx (HIP).  Differences, all additive:

* ``singleImageInferenceAll(image, mode)`` returns every class plane from ONE pass over the slide (the reference
  re-runs the whole slide per class, UnMicst1-5.py:847-848).  ``singleImageInference`` is served from that pass: a
  repeated call with the same image object and mode reuses it (``UNet2D.reuse_pass = False`` switches that off).
* ``gpuIndex == -1`` picks the device with the most free HBM (the reference asks NVML, toolbox/GPUselect.py:4-22).
* errors are Python exceptions like the reference's: ``FileNotFoundError``/``KeyError`` for a model directory that
  does not match the graph (TensorFlow raises NotFoundError on restore), ``umx.UmxError`` for device failures.
  There is no CPU fallback.
"""
from __future__ import annotations

import os

import numpy as np

from . import model as _model
from . import umx as _umx


class UNet2D:
    hp = None               # the reference exposes the hp dict here (UNet2D.hp['nClasses'], UnMicst1-5.py:771)
    hparams = None          # typed view of the same
    DatasetMean = None
    DatasetStDev = None
    Engine = None           # umx.Engine (the reference keeps its tf.Session in UNet2D.Session)
    reuse_pass = True
    stitch = _umx.STITCH_FP16_COMPAT
    max_batch = 0           # 0 = chosen from the tile size
    _last = None            # (key, planes) of the most recent full pass

    # ---------------------------------------------------------------- training (v2 graph)

    @staticmethod
    def setupWithHP(hp):
        """== reference UnMicst1-5.py:42-53."""
        UNet2D.setup(hp["imSize"], hp["nChannels"], hp["nClasses"], hp["nOut0"], hp["featMapsFact"], hp["downSampFact"],
                     hp["ks"], hp["nExtraConvs"], hp["stdDev0"], hp["nLayers"], hp["batchSize"])

    @staticmethod
    def setup(imSize, nChannels, nClasses, nOut0, featMapsFact, downSampFact, kernelSize, nExtraConvs, stdDev0,
              nDownSampLayers, batchSize):
        """== reference UnMicst1-5.py:55-68: records the hyper-parameters (the graph itself is built by libumx when an
        engine or a trainer is created from them)."""
        UNet2D.hp = {"imSize": imSize, "nClasses": nClasses, "nChannels": nChannels, "nExtraConvs": nExtraConvs,
                     "nLayers": nDownSampLayers, "featMapsFact": featMapsFact, "downSampFact": downSampFact,
                     "ks": kernelSize, "nOut0": nOut0, "stdDev0": stdDev0, "batchSize": batchSize}
        UNet2D.hparams = _model.hparams_from_dict(UNet2D.hp, _model.GRAPH_V2)

    @staticmethod
    def train(*args, **kwargs):
        """Out of scope: the reference's data loading, augmentation and bookkeeping around the
        optimisation step (UnMicst1-5.py:240-578) are not part of the path this package replaces.  The step itself is
        ``unmicst_amd.trainer.Trainer.step`` (include/umx_train.h); INTEGRATION.md shows where it goes in the reference's loop."""
        raise NotImplementedError("UNet2D.train is out of scope: drive unmicst_amd.trainer.Trainer.step from your own loop "
                                  "(INTEGRATION.md, 'Training step')")

    @staticmethod
    def deploy(*args, **kwargs):
        """Out of scope like ``train`` (reference UnMicst1-5.py:583-654); ``singleImageInference`` is the inference entry."""
        raise NotImplementedError("UNet2D.deploy is out of scope: use UNet2D.singleImageInference")

    # ---------------------------------------------------------------- setup / cleanup
    @staticmethod
    def singleImageInferenceSetup(modelPath, gpuIndex, mean, std, graph=None):
        """== reference UnMicst1-5.py:656-682.  ``graph`` None: detected from the checkpoint's variable names."""
        synthetic = os.environ.get("UMX_SYNTHETIC_WEIGHTS", "0") not in ("", "0")
        art = _model.load_model_dir(modelPath, graph, synthetic_if_missing=synthetic)
        UNet2D.setupWithArtefacts(art, gpuIndex, mean, std)

    @staticmethod
    def setupWithArtefacts(art, gpuIndex=-1, mean=-1, std=-1):
        UNet2D.singleImageInferenceCleanup()
        hp = art.hp
        UNet2D.hparams = hp
        UNet2D.hp = {"imSize": hp.imSize, "nClasses": hp.nClasses, "nChannels": hp.nChannels,
                     "nExtraConvs": hp.nExtraConvs, "nLayers": hp.nLayers, "featMapsFact": hp.featMapsFact,
                     "downSampFact": hp.downSampFact, "ks": hp.ks, "nOut0": hp.nOut0, "batchSize": hp.batchSize}
        UNet2D.DatasetMean = art.mean if mean == -1 else mean
        UNet2D.DatasetStDev = art.std if std == -1 else std
        print(UNet2D.DatasetMean)
        print(UNet2D.DatasetStDev)
        device = _umx.pick_device_most_free_memory() if gpuIndex is None or gpuIndex < 0 else int(gpuIndex)
        # tiles per launch group: enough M to fill 256 CUs at the deepest level (a few hundred MB of activations)
        batch = UNet2D.max_batch or _umx.auto_batch(hp)
        UNet2D._engine_args = (hp, art.blob, device, batch)
        UNet2D.Engine = _umx.Engine(hp, art.blob, device=device, max_batch=batch)
        print("Model restored.")

    @staticmethod
    def singleImageInferenceCleanup():
        """== reference UnMicst1-5.py:684-685."""
        if UNet2D.Engine is not None:
            UNet2D.Engine.close()
        UNet2D.Engine = None
        UNet2D._last = None

    # ---------------------------------------------------------------- inference
    @staticmethod
    def _with_range_fallback(call):
        """Run ``call()`` on the engine; if the split-precision path reports UMX_ERR_RANGE (an activation left binary16's
        range -- the reference's fp32 TensorFlow graph has no such limit), rebuild the engine with the exact-fp32 MFMA
        kernels in this process and run the call again, so the CLI succeeds wherever the reference does."""
        try:
            return call()
        except _umx.UmxError as e:
            if e.code != _umx.ERR_RANGE or UNet2D.Engine is None or UNet2D.Engine.precision == "f32":
                raise
            hp, blob, device, batch = UNet2D._engine_args
            print("split-precision range exceeded: switching this model to the exact-fp32 kernels")
            UNet2D.Engine.close()
            UNet2D.Engine = _umx.Engine(hp, blob, device=device, max_batch=batch, precision="f32")
            return call()

    @staticmethod
    def _check_image(image):
        hp = UNet2D.hparams
        image = np.asarray(image)
        if image.ndim == 2:
            return image
        if image.ndim == 3:
            # duo feeds channel c of the tile from plane c (UnMicst2.py:679-681); solo copies the one 2-D patch to
            # every channel (UnMicst1-5.py:701-702) -- a 3-D image there would not broadcast, so reject the mismatch
            if image.shape[0] != hp.nChannels:
                raise ValueError("image has %d planes, the model takes %d channels" % (image.shape[0], hp.nChannels))
            return image
        raise ValueError("image must be (H, W) or channel-first (C, H, W)")

    @staticmethod
    def singleImageInferenceAll(image, mode="accumulate"):
        """All class planes [nClasses, H, W] (float16; float32 if ``UNet2D.stitch`` is STITCH_FP32) in one pass."""
        if UNet2D.Engine is None:
            raise RuntimeError("call UNet2D.singleImageInferenceSetup first")
        if mode not in ("accumulate", "replace"):
            raise ValueError("mode must be 'accumulate' or 'replace'")  # PI2D.Mode, PartitionOfImage.py:75
        print("Inference...")
        image = UNet2D._check_image(image)
        m = _umx.MODE_ACCUMULATE if mode == "accumulate" else _umx.MODE_REPLACE
        return UNet2D._with_range_fallback(
            lambda: UNet2D.Engine.infer_image(image, UNet2D.DatasetMean, UNet2D.DatasetStDev, m, UNet2D.stitch))

    @staticmethod
    def singleImageInferenceRaw(raw, rescale, mode="accumulate", value_range=None):
        """Driver fast path (``--scalingFactor 1``, ``--outlier -1``): raw uint8/uint16 plane(s) -> uint8 class planes
        [nClasses, H, W], with the drivers' im2double / rescale_intensity / double uint8 cast done on the GPU
        (reference UnMicst1-5.py:807-821,848-854).  ``rescale`` False reproduces solo's un-rescaled input.  ``value_range``: per
        plane (min, max) of the raw samples where the caller has them already (the upload then overlaps the inference)."""
        if UNet2D.Engine is None:
            raise RuntimeError("call UNet2D.singleImageInferenceSetup first")
        if mode not in ("accumulate", "replace"):
            raise ValueError("mode must be 'accumulate' or 'replace'")
        print("Inference...")
        raw = np.asarray(raw)
        if raw.ndim == 3 and raw.shape[0] != UNet2D.hparams.nChannels:
            raise ValueError("image has %d planes, the model takes %d channels" % (raw.shape[0], UNet2D.hparams.nChannels))
        m = _umx.MODE_ACCUMULATE if mode == "accumulate" else _umx.MODE_REPLACE
        return UNet2D._with_range_fallback(
            lambda: UNet2D.Engine.infer_image_raw(raw, bool(rescale), UNet2D.DatasetMean, UNet2D.DatasetStDev, m,
                                                  value_range=value_range if rescale else None))

    @staticmethod
    def singleImageInferenceRawScaled(raw, scaling, rescale, mode="accumulate"):
        """Driver fast path at ``--scalingFactor != 1`` (``--outlier -1``): resize -> [rescale] -> inference -> uint8 -> resize
        back, all on the device (reference UnMicst1-5.py:813-821,848-854)."""
        if UNet2D.Engine is None:
            raise RuntimeError("call UNet2D.singleImageInferenceSetup first")
        print("Inference...")
        raw = np.asarray(raw)
        if raw.ndim == 3 and raw.shape[0] != UNet2D.hparams.nChannels:
            raise ValueError("image has %d planes, the model takes %d channels" % (raw.shape[0], UNet2D.hparams.nChannels))
        m = _umx.MODE_ACCUMULATE if mode == "accumulate" else _umx.MODE_REPLACE
        return UNet2D._with_range_fallback(
            lambda: UNet2D.Engine.infer_image_raw_scaled(raw, float(scaling), bool(rescale), UNet2D.DatasetMean,
                                                         UNet2D.DatasetStDev, m))

    @staticmethod
    def singleImageInferenceRawOutlier(raw, scaling, outlier, mode="accumulate"):
        """Driver fast path with ``--outlier``: resize -> rescale to (min, percentile) -> inference -> uint8 -> resize back, all on
        the device (reference UnMicst1-5.py:813-821,848-854; the percentile is numpy's, found by radix selection)."""
        if UNet2D.Engine is None:
            raise RuntimeError("call UNet2D.singleImageInferenceSetup first")
        print("Inference...")
        raw = np.asarray(raw)
        if raw.ndim == 3 and raw.shape[0] != UNet2D.hparams.nChannels:
            raise ValueError("image has %d planes, the model takes %d channels" % (raw.shape[0], UNet2D.hparams.nChannels))
        m = _umx.MODE_ACCUMULATE if mode == "accumulate" else _umx.MODE_REPLACE
        return UNet2D._with_range_fallback(
            lambda: UNet2D.Engine.infer_image_raw_outlier(raw, float(scaling), float(outlier), UNet2D.DatasetMean,
                                                          UNet2D.DatasetStDev, m))

    @staticmethod
    def _pass_key(image, mode):
        a = np.asarray(image)
        flat = a.reshape(-1)
        step = max(1, flat.size // 4099)
        return (id(image), a.__array_interface__["data"][0], a.shape, str(a.dtype), mode, UNet2D.stitch,
                float(UNet2D.DatasetMean), float(UNet2D.DatasetStDev), flat[::step].tobytes())

    @staticmethod
    def singleImageInference(image, mode, pmIndex):
        """== reference UnMicst1-5.py:687-710: the stitched probability plane of class ``pmIndex``."""
        K = UNet2D.hparams.nClasses if UNet2D.hparams else 0
        if not 0 <= int(pmIndex) < K:
            raise IndexError("pmIndex %r out of range for %d classes" % (pmIndex, K))
        key = UNet2D._pass_key(image, mode) if UNet2D.reuse_pass else None
        if key is not None and UNet2D._last is not None and UNet2D._last[0] == key:
            planes = UNet2D._last[1]
        else:
            planes = UNet2D.singleImageInferenceAll(image, mode)
            UNet2D._last = (key, planes) if key is not None else None
        return planes[int(pmIndex)]
