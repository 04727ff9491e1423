"""``UNet2D.train`` -- host-side mirror of the reference's training function, driving the HIP training step.

Reference: ``UNet2D.train(imPath, validPath, testPath, logPath, modelPath, pmPath, nTrain, nValid, nTest,
restoreVariables, nSteps, gpuIndex, testPMIndex)`` (UnMicst1-5.py:240-578 solo, UnMicst2.py:237-561 duo).  Kept: the
argument list; the data convention (``I%05d_Img.tif`` pages, ``_Ant.tif`` class ids 1..K, ``_wt.tif`` contour weights;
class weights bg / contour + k*W / nuclei); normalisation with the script's hard-coded dataset mean / st.dev and the two
pickles it saves; the brightness / contrast jitter; the epoch bookkeeping; a validation batch after every training batch
in inference mode with the per-class pixel error; "save when step % 50 == 0 and the error improved"; ``hp.data``; the test
pass that writes ``[sqrt(normalised input) | probability | ground truth]`` PNGs.  The per-script differences are data
(``RECIPES``), including the scripts' quirks (solo validates on channel 0 only; duo as shipped reads page 0 into
channel 0 only and has no augmentation dimension).

Replaced: the TensorFlow graph, optimiser and session (-> ``trainer.Trainer``: libumx, HIP); TensorBoard summaries
(-> ``<logPath>/Train|Valid/scalars.csv``); ``saver.save`` (-> ``umx_model.npz``, the blob the inference engine loads).
There is no CPU fallback: the step runs on the GPU or raises.
"""
from __future__ import annotations

import math
import os
import pickle
import shutil
import struct
import zlib
from dataclasses import dataclass
from typing import Callable, Optional

import numpy as np

from . import imtools, model, tiffio
from . import trainer as _trainer


@dataclass(frozen=True)
class Recipe:
    script: str
    n_aug: int                 # augmentation pages per channel in I%05d_Img.tif (0: no augmentation dimension)
    all_channels: bool         # read every channel's pages (solo) or page 0 into channel 0 only (duo as shipped)
    dataset_mean: float
    dataset_stdev: float
    bg_weight: float
    contour_weight: float
    nuclei_weight: float
    intersect_weight: float
    valid_channel0_only: bool  # solo's validation / test batches fill channel 0 only (UnMicst1-5.py:493-495,556-557)
    summary_every: int
    png_aug_suffix: bool       # solo: I%05d_%d_Nuc.png per augmentation; duo: I%05dNuc.png
    options: Callable[[], _trainer.TrainOptions]


RECIPES = {
    # UnMicst1-5.py:258,277-282,464-465,489,493-495,572
    "solo": Recipe("UnMicst1-5.py", 12, True, 0.34, 0.25, 1, 2, 7, 15, True, 20, True, _trainer.solo_options),
    # UnMicst2.py:273-278,295-296,478,481-483,553
    "duo": Recipe("UnMicst2.py", 0, False, 0.19, 0.17, 1, 2, 5, 10, False, 1, False, _trainer.duo_options),
}


def save_data(data, path):            # toolbox/ftools.py:32-35
    print("saving data")
    with open(path, "wb") as f:
        pickle.dump(data, f)


def normalize(I):                     # toolbox/imtools.py:70-76
    m, M = np.min(I), np.max(I)
    return (I - m) / (M - m) if M > m else I


def png_write(path: str, img_u8: np.ndarray) -> None:
    """8-bit grayscale PNG (skimage.io.imsave of a uint8 2-D array, toolbox/imtools.py:39-40)."""
    img = np.ascontiguousarray(img_u8, dtype=np.uint8)
    h, w = img.shape
    raw = b"".join(b"\x00" + img[r].tobytes() for r in range(h))

    def chunk(tag, payload):
        c = struct.pack(">I", len(payload)) + tag + payload
        return c + struct.pack(">I", zlib.crc32(tag + payload) & 0xFFFFFFFF)

    with open(path, "wb") as f:
        f.write(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 0, 0, 0, 0)) +
                chunk(b"IDAT", zlib.compress(raw, 6)) + chunk(b"IEND", b""))


def initial_blob(hp: model.HParams, std_dev0: float, rng: np.random.Generator) -> np.ndarray:
    """``tf.global_variables_initializer()`` of the v2 graph: kernelD%d ~ truncated_normal(stddev=stdDev0)
    (UnMicst1-5.py:85-87); every other kernel VarianceScaling(fan_in) = truncated normal with
    stddev sqrt(1/fan_in)/0.87962566 (:93,105,126,162,166,214); BN gamma 1, beta 0, moving mean 0, variance 1."""
    def trunc(shape, std):
        x = rng.normal(0.0, 1.0, shape)
        bad = np.abs(x) > 2.0
        while bad.any():
            x[bad] = rng.normal(0.0, 1.0, int(bad.sum()))
            bad = np.abs(x) > 2.0
        return x * std

    tensors = {}
    for name, shape in model.tensor_specs(hp):
        if name.endswith(".bn.gamma") or name.endswith(".bn.var"):
            t = np.ones(shape)
        elif name.endswith(".bn.beta") or name.endswith(".bn.mean"):
            t = np.zeros(shape)
        elif name.endswith(".w1"):
            t = trunc(shape, std_dev0)
        else:
            fan_in = int(np.prod(shape[:-2])) * shape[-2]      # keras: receptive field x shape[-2]
            t = trunc(shape, math.sqrt(1.0 / fan_in) / 0.87962566103423978)
        tensors[name] = t.astype(np.float32)
    return model.blob_from_tensors(hp, tensors)


def load_split(path: str, n: int, order, hp: model.HParams, rec: Recipe):
    """-> (images [n,P,P,nAug|1,C] normalised, labels [n,P,P,K], weights [n,P,P,K])  (UnMicst1-5.py:295-312)."""
    P, C, K = hp.imSize, hp.nChannels, hp.nClasses
    A = max(rec.n_aug, 1)
    X = np.zeros((n, P, P, A, C))
    L = np.zeros((n, P, P, K))
    W = np.zeros((n, P, P, K))
    for i in range(n):
        src = int(order[i])
        img_path = "%s/I%05d_Img.tif" % (path, src)
        if rec.all_channels:
            for c in range(C):
                for a in range(A):
                    X[i, :, :, a, c] = (imtools.im2double(tiffio.imread(img_path, key=a + A * c)) - rec.dataset_mean) / rec.dataset_stdev
        else:
            X[i, :, :, 0, 0] = (imtools.im2double(tiffio.imread(img_path, key=0)) - rec.dataset_mean) / rec.dataset_stdev
        ant = tiffio.imread("%s/I%05d_Ant.tif" % (path, src))
        wt = tiffio.imread("%s/I%05d_wt.tif" % (path, src))
        for k in range(K):
            L[i, :, :, k] = (ant == k + 1)
            if k == 1:
                W[i, :, :, k] = wt * rec.intersect_weight + rec.contour_weight
            elif k == 2:
                W[i, :, :, k] = wt * 0 + rec.nuclei_weight
            else:
                W[i, :, :, k] = wt * 0 + rec.bg_weight
    return X, L, W


def pixel_errors(probs: np.ndarray, labels: np.ndarray) -> np.ndarray:
    """Per class: 1 - |label == k and argmax == k| / |label == k|   (UnMicst1-5.py:386-397)."""
    pred = np.argmax(probs, axis=3)
    out = np.zeros(labels.shape[3])
    with np.errstate(divide="ignore", invalid="ignore"):
        for k in range(labels.shape[3]):
            lab = labels[..., k].astype(np.int32)
            out[k] = 1.0 - np.float32(np.sum(lab * (pred == k))) / np.float32(np.sum(lab))
    return out


class _Scalars:
    def __init__(self, directory: str, n_classes: int):
        os.makedirs(directory, exist_ok=True)
        self.f = open(os.path.join(directory, "scalars.csv"), "w")
        self.f.write("step,avg_cross_entropy," + ",".join("avg_pixel_error_%d" % k for k in range(n_classes)) + ",learning_rate\n")

    def add(self, step, loss, errors, lr):
        self.f.write("%d,%.9g,%s,%.9g\n" % (step, loss, ",".join("%.6g" % e for e in errors), lr))
        self.f.flush()

    def close(self):
        self.f.close()


def train(hp_dict: dict, imPath, validPath, testPath, logPath, modelPath, pmPath, nTrain, nValid, nTest,
          restoreVariables, nSteps, gpuIndex, testPMIndex, regime: str = "solo",
          trainer_factory: Optional[Callable] = None):
    """The reference's ``UNet2D.train`` with ``UNet2D.hp`` passed in.  ``trainer_factory(hp, blob, opts, batch, device)``
    defaults to the HIP ``trainer.Trainer``.  Returns the list of per-step (loss, mean validation error)."""
    rec = RECIPES[regime]
    hp = model.hparams_from_dict(hp_dict, model.GRAPH_V2)
    make = trainer_factory or (lambda hp_, blob_, opts_, batch_, device_: _trainer.Trainer(hp_, blob_, opts_, batch=batch_,
                                                                                          device=device_))
    B, P, C, K = hp.batchSize, hp.imSize, hp.nChannels, hp.nClasses
    A = max(rec.n_aug, 1)

    print("loading data, computing mean / st dev")
    os.makedirs(modelPath, exist_ok=True)
    save_data(rec.dataset_mean, os.path.join(modelPath, "datasetMean.data"))
    save_data(rec.dataset_stdev, os.path.join(modelPath, "datasetStDev.data"))
    perm = np.arange(nTrain)
    np.random.shuffle(perm)
    Train, LTrain, WTrain = load_split(imPath, nTrain, perm, hp, rec)
    valid_order = np.arange(nValid)
    np.random.shuffle(valid_order)
    Valid, LValid, WValid = load_split(validPath, nValid, valid_order, hp, rec)
    Test, LTest, _ = load_split(testPath, nTest, np.arange(nTest), hp, rec)

    opts = rec.options()
    if restoreVariables:
        blob = model.load_model_dir(modelPath, model.GRAPH_V2).blob
        print("Model restored.")
    else:
        blob = initial_blob(hp, float(hp_dict.get("stdDev0", 0.03)), np.random.default_rng(np.random.randint(1 << 31)))
    tr = make(hp, blob, opts, B, max(int(gpuIndex), 0))

    if os.path.exists(logPath):
        shutil.rmtree(logPath)
    train_log = _Scalars(os.path.join(logPath, "Train"), K)
    valid_log = _Scalars(os.path.join(logPath, "Valid"), K)

    x_batch = np.zeros((B, P, P, C))
    y_batch = np.zeros((B, P, P, K))
    w_batch = np.zeros((B, P, P, K))
    train_order = np.arange(nTrain)
    np.random.shuffle(train_order)
    valid_order = np.arange(nValid)
    np.random.shuffle(valid_order)
    brightness_span = 1 * rec.dataset_stdev
    contrast_span = 0.1 * rec.dataset_stdev
    jT = jV = 0
    epoch = 1
    best_error = np.inf
    history = []
    saved = False
    try:
        for i in range(nSteps):
            for j in range(B):
                # (brightness shift and contrast gain of one image: a random sign times a uniform fraction of the span,
                # the reference's augmentation UnMicst1-5.py:472-475)
                shift = (-1.0 if np.random.rand() < 0.5 else 1.0) * brightness_span * np.random.rand()
                gain = 1.0 + (-1.0 if np.random.rand() < 0.5 else 1.0) * contrast_span * np.random.rand()
                if rec.all_channels:
                    for c in range(C):
                        x_batch[j, :, :, c] = Train[train_order[jT + j], :, :, math.floor(A * np.random.rand()), c] * gain + shift
                else:
                    x_batch[j] = Train[train_order[jT + j], :, :, 0, :] * gain + shift
                y_batch[j] = LTrain[train_order[jT + j]]
                w_batch[j] = WTrain[train_order[jT + j]]
            lr = opts.lr0 * opts.decay_rate ** (tr.step_count // opts.decay_steps)
            loss = tr.step(x_batch, y_batch, w_batch)[0]
            jT += B
            if jT > (nTrain - B - 1):
                jT = 0
                np.random.shuffle(train_order)
                epoch += 1
            if i % rec.summary_every == 0:
                train_log.add(i, loss, pixel_errors(tr.probs(), y_batch), lr)

            for j in range(B):
                if rec.valid_channel0_only:
                    x_batch[j] = 0
                    x_batch[j, :, :, 0] = Valid[valid_order[jV + j], :, :, math.floor(A * np.random.rand()), 0]
                else:
                    x_batch[j] = Valid[valid_order[jV + j], :, :, 0, :]
                y_batch[j] = LValid[valid_order[jV + j]]
                w_batch[j] = WValid[valid_order[jV + j]]
            es = pixel_errors(tr.eval(x_batch), y_batch)
            jV += B
            if jV > (nValid - B - 1):
                jV = 0
                np.random.shuffle(valid_order)
            if i % rec.summary_every == 0:
                valid_log.add(i, float("nan"), es, lr)
            e = float(np.mean(es))
            print("step %05d, e: %f" % (i, e) + ", epoch: " + str(epoch))
            history.append((loss, e))
            if i == 0:
                best_error = e if restoreVariables else np.inf
            if i % 50 == 0 and e < best_error:
                best_error = e
                out = model.save_converted(model.ModelArtefacts(hp, tr.blob(), rec.dataset_mean, rec.dataset_stdev), modelPath)
                saved = True
                print("Model saved in file: %s" % out)
        save_data(dict(hp_dict), os.path.join(modelPath, "hp.data"))
    finally:
        train_log.close()
        valid_log.close()
        tr.close()

    # ---- test (UnMicst1-5.py:533-578): restore the saved model, write [input | probability | ground truth] PNGs
    if nTest > 0 and saved:
        art = model.load_model_dir(modelPath, model.GRAPH_V2)
        print("Model restored.")
        te = make(hp, art.blob, opts, B, max(int(gpuIndex), 0))
        os.makedirs(pmPath, exist_ok=True)
        try:
            for a in range(A if rec.png_aug_suffix else 1):
                for i in range(nTest):
                    j = i % B
                    if rec.valid_channel0_only:
                        x_batch[j] = 0
                        x_batch[j, :, :, 0] = Test[i, :, :, a, 0]
                    else:
                        x_batch[j] = Test[i, :, :, 0, :]
                    y_batch[j] = LTest[i]
                    if j == B - 1 or i == nTest - 1:
                        output = te.eval(x_batch)
                        for k in range(j + 1):
                            im = np.sqrt(normalize(x_batch[k, :, :, 0]))
                            for cls, tag in ((2, "Nuc"), (1, "Con")):
                                pane = np.concatenate((im, np.concatenate((output[k, :, :, cls], y_batch[k, :, :, cls]), axis=1)), axis=1)
                                name = ("I%05d_%d_%s.png" % (i - j + k + 1, a, tag)) if rec.png_aug_suffix else \
                                    ("I%05d%s.png" % (i - j + k + 1, tag))
                                png_write(os.path.join(pmPath, name), np.uint8(255 * pane))
        finally:
            te.close()
    return history


def deploy(imPath, nImages, modelPath, pmPath, gpuIndex, pmIndex, engine_factory: Optional[Callable] = None):
    """The reference's ``UNet2D.deploy`` (UnMicst1-5.py:583-654): images ``I%05d_Img.tif`` of the model's tile size, one
    page per channel, normalised with the model's mean / st.dev, pushed through the network in batches; writes
    ``I%05d_Im.png`` (sqrt of the min-max normalised first channel) and ``I%05d_PM.png`` (class ``pmIndex``).
    ``engine_factory(hp, blob, device, max_batch)`` defaults to the HIP inference engine."""
    from . import umx as _umx
    art = model.load_model_dir(modelPath)
    hp = art.hp
    B, P, C = hp.batchSize, hp.imSize, hp.nChannels
    make = engine_factory or (lambda hp_, blob_, device_, mb_: _umx.Engine(hp_, blob_, device=device_, max_batch=mb_))
    Data = np.zeros((nImages, P, P, C))
    for i in range(nImages):
        path = "%s/I%05d_Img.tif" % (imPath, i)
        for c in range(C):
            Data[i, :, :, c] = (imtools.im2double(tiffio.imread(path, key=c)) - art.mean) / art.std
    eng = make(hp, art.blob, max(int(gpuIndex), 0), B)
    print("Model restored.")
    os.makedirs(pmPath, exist_ok=True)
    x_batch = np.zeros((B, P, P, C))
    try:
        for i in range(nImages):
            print(i, nImages)
            j = i % B
            x_batch[j] = Data[i]
            if j == B - 1 or i == nImages - 1:
                output = eng.forward_tiles(x_batch[:j + 1].astype(np.float32))
                for k in range(j + 1):
                    im = np.sqrt(normalize(x_batch[k, :, :, 0]))
                    png_write("%s/I%05d_Im.png" % (pmPath, i - j + k + 1), np.uint8(255 * im))
                    png_write("%s/I%05d_PM.png" % (pmPath, i - j + k + 1), np.uint8(255 * output[k, :, :, pmIndex]))
    finally:
        eng.close()
