"""Model artefacts: hyper-parameters, checkpoint -> canonical weight blob, seeded synthetic weights.

Mirrors the reference's artefact handling without TensorFlow:

* ``hp.data`` / ``datasetMean.data`` / ``datasetStDev.data`` are plain pickles read with
  ``toolbox.ftools.loadData`` (reference toolbox/ftools.py:37-40, used at UnMicst1-5.py:659-669).
* weights are restored from ``model.ckpt`` (reference UnMicst1-5.py:677-681) -- here through
  :mod:`unmicst_amd.tfckpt`.

The *canonical weight blob* is one flat little-endian float32 array holding the raw TensorFlow tensors
in graph-execution order (documented in include/umx.h).  All folding (BatchNorm -> scale/bias,
main+shortcut filter sum, transposed-conv phase split, MFMA packing) happens inside the C library at
``umx_create`` time, so the blob is a pure data artefact.
"""
from __future__ import annotations

import os
import pickle
from dataclasses import dataclass
from typing import Dict, List, Tuple

import numpy as np

from . import tfckpt

GRAPH_LEGACY = 0  # reference UnMicst.py:51-187
GRAPH_V2 = 1      # reference UnMicst1-5.py:55-237 / UnMicst2.py:52-235


@dataclass(frozen=True)
class HParams:
    """The reference's ``hp`` dict (UnMicst1-5.py:57-67) plus which graph builder consumes it."""
    graph: int
    imSize: int
    nChannels: int
    nClasses: int
    nOut0: int
    nLayers: int
    ks: int
    nExtraConvs: int
    featMapsFact: int = 2
    downSampFact: int = 2
    batchSize: int = 16

    @property
    def nOutX(self) -> List[int]:
        n = [self.nChannels, self.nOut0]
        for _ in range(self.nLayers):
            n.append(n[-1] * self.featMapsFact)
        return n

    @property
    def margin(self) -> int:
        return int(self.imSize / 8)  # reference UnMicst1-5.py:694

    def validate(self) -> None:
        if self.graph not in (GRAPH_LEGACY, GRAPH_V2):
            raise ValueError("unknown graph kind %r" % (self.graph,))
        if self.downSampFact != 2 or self.featMapsFact < 1:
            raise ValueError("only downSampFact == 2 is supported (every shipped model uses 2)")
        if self.ks % 2 != 1 or self.ks < 1 or self.ks > 7:
            raise ValueError("kernel size must be odd and <= 7")
        if self.imSize % (1 << self.nLayers) != 0 or (self.imSize >> self.nLayers) < 1:
            raise ValueError("imSize must be divisible by 2**nLayers")
        if self.imSize % 8 != 0:
            raise ValueError("imSize must be a multiple of 8 (margin = imSize/8)")

    def flops_per_tile(self) -> float:
        """Algorithmic FLOPs (2*MAC) of one tile forward, graph as written."""
        n = self.nOutX
        ks2 = self.ks * self.ks
        kss2 = ks2 if self.graph == GRAPH_V2 else 1
        mac = 0
        s = self.imSize
        for i in range(self.nLayers):
            px = s * s
            mac += px * ks2 * n[i] * n[i + 1]
            mac += px * ks2 * n[i + 1] * n[i + 1] * self.nExtraConvs
            mac += px * kss2 * n[i] * n[i + 1]
            s //= 2
        mac += s * s * ks2 * n[self.nLayers] * n[self.nLayers + 1]
        for idx in range(self.nLayers - 1, -1, -1):
            mac += s * s * ks2 * n[idx + 2] * n[idx + 1]  # transposed conv, dense count on the input grid
            s *= 2
            px = s * s
            mac += px * ks2 * (n[idx] + n[idx + 1]) * n[idx + 1]
            mac += px * ks2 * n[idx + 1] * n[idx + 1] * self.nExtraConvs
        mac += s * s * n[1] * self.nClasses
        return 2.0 * mac


def load_pickle(path: str):
    with open(path, "rb") as f:
        return pickle.load(f)


def hparams_from_dict(hp: dict, graph: int) -> HParams:
    h = HParams(graph=graph, imSize=int(hp["imSize"]), nChannels=int(hp["nChannels"]),
                nClasses=int(hp["nClasses"]), nOut0=int(hp["nOut0"]), nLayers=int(hp["nLayers"]),
                ks=int(hp["ks"]), nExtraConvs=int(hp["nExtraConvs"]),
                featMapsFact=int(hp.get("featMapsFact", 2)), downSampFact=int(hp.get("downSampFact", 2)),
                batchSize=int(hp.get("batchSize", 16)))
    h.validate()
    return h


def tensor_specs(hp: HParams) -> List[Tuple[str, Tuple[int, ...]]]:
    """Canonical blob order: (name, shape) of every tensor, graph-execution order."""
    n = hp.nOutX
    ks = hp.ks
    v2 = hp.graph == GRAPH_V2
    kss = ks if v2 else 1
    specs: List[Tuple[str, Tuple[int, ...]]] = []

    def bn(prefix, c):
        for t in ("gamma", "beta", "mean", "var"):
            specs.append(("%s.bn.%s" % (prefix, t), (c,)))

    for i in range(hp.nLayers):
        specs.append(("ld%d.w1" % i, (ks, ks, n[i], n[i + 1])))
        for e in range(hp.nExtraConvs):
            specs.append(("ld%d.wextra%d" % (i, e), (ks, ks, n[i + 1], n[i + 1])))
        specs.append(("ld%d.wshort" % i, (kss, kss, n[i], n[i + 1])))
        bn("ld%d" % i, n[i + 1])
    specs.append(("lb.w", (ks, ks, n[hp.nLayers], n[hp.nLayers + 1])))
    if v2:
        bn("lb", n[hp.nLayers + 1])
    for idx in range(hp.nLayers - 1, -1, -1):
        specs.append(("lu%d.wt" % idx, (ks, ks, n[idx + 1], n[idx + 2])))
        specs.append(("lu%d.w2" % idx, (ks, ks, n[idx] + n[idx + 1], n[idx + 1])))
        if v2:
            bn("lu%d" % idx, n[idx + 1])
        for e in range(hp.nExtraConvs):
            specs.append(("lu%d.wextra%d" % (idx, e), (ks, ks, n[idx + 1], n[idx + 1])))
    specs.append(("lt.w", (1, 1, n[1], hp.nClasses)))
    if v2:
        bn("lt", hp.nClasses)
    return specs


def _ckpt_name(hp: HParams, canon: str) -> str:
    """Canonical tensor name -> TensorFlow variable name in the reference's checkpoints."""
    layer, rest = canon.split(".", 1)
    bn_map = {"bn.gamma": "gamma", "bn.beta": "beta", "bn.mean": "moving_mean", "bn.var": "moving_variance"}
    if hp.graph == GRAPH_LEGACY:
        # names from reference UnMicst.py:81-96,108-111,134-145,167-168 (tf.name_scope prefixes)
        if layer.startswith("ld"):
            i = int(layer[2:])
            if rest in bn_map:
                return ("batch_normalization/" if i == 0 else "batch_normalization_%d/" % i) + bn_map[rest]
            sub = {"w1": "kernel1", "wshort": "shortcutWeights"}.get(rest) or rest.replace("wextra", "kernelExtra")
            return "downsampling/ld%d/%s" % (i, sub)
        if layer == "lb":
            return "lb/kernel1"
        if layer.startswith("lu"):
            i = int(layer[2:])
            sub = {"wt": "kernel1", "w2": "kernel2"}.get(rest) or rest.replace("wextra", "kernel2Extra")
            return "upsampling/lu%d/%s" % (i, sub)
        if layer == "lt":
            return "lt/kernel"
    else:
        # names from reference UnMicst1-5.py:85-109,124-138,159-174,212-222 (variable_scope / name_scope mix)
        if layer.startswith("ld"):
            i = int(layer[2:])
            if rest in bn_map:
                return "ld%d/batch_normalization/%s" % (i, bn_map[rest])
            if rest == "w1":
                return "downsampling/ld%d/kernelD%d" % (i, i)
            if rest == "wshort":
                return "ld%d/shortcutWeights" % i
            return "ld%d/%s" % (i, rest.replace("wextra", "kernelExtra"))
        if layer == "lb":
            return "lb/kernel1" if rest == "w" else "conv/" + bn_map[rest]
        if layer.startswith("lu"):
            i = int(layer[2:])
            if rest in bn_map:
                return "lu%d/conv2/%s" % (i, bn_map[rest])
            if rest == "wt":
                return "lu%d/kernelU%d" % (i, i)
            if rest == "w2":
                return "lu%d/kernel2" % i
            return "lu%d/%s" % (i, rest.replace("wextra", "kernel2Extra"))
        if layer == "lt":
            return "lt/kernel" if rest == "w" else "batch_normalization/" + bn_map[rest]
    raise KeyError(canon)


def blob_from_tensors(hp: HParams, tensors: Dict[str, np.ndarray]) -> np.ndarray:
    parts = []
    for name, shape in tensor_specs(hp):
        t = np.asarray(tensors[name], dtype=np.float32)
        if tuple(t.shape) != tuple(shape):
            raise ValueError("tensor %s has shape %s, expected %s" % (name, t.shape, shape))
        parts.append(t.ravel())
    return np.ascontiguousarray(np.concatenate(parts), dtype="<f4")


def tensors_from_blob(hp: HParams, blob: np.ndarray) -> Dict[str, np.ndarray]:
    out = {}
    pos = 0
    for name, shape in tensor_specs(hp):
        n = int(np.prod(shape))
        out[name] = blob[pos:pos + n].reshape(shape)
        pos += n
    if pos != blob.size:
        raise ValueError("blob has %d floats, graph needs %d" % (blob.size, pos))
    return out


def hp_dict_from_checkpoint(hp: dict, names: Dict[str, dict]) -> dict:
    """``hp.data`` reconciled with the tensor shapes of the checkpoint that is actually loaded.  The reference's
    ``models/mousenucleiDAPI`` ships the hp.data of a 20-feature model whose shard is missing (.MISSING_LARGE_BLOBS:3) next
    to ``nuclei20x2bin1chan.*`` (16 features): widths, kernel size, depth, extra convolutions, channels and classes are read
    off the variables (names as in reference UnMicst.py:81-96,167-168 / UnMicst1-5.py:85-109,212-222); everything else
    (imSize, batchSize, ...) stays as hp.data has it."""
    out = dict(hp)
    first = names.get("downsampling/ld0/kernel1") or names.get("downsampling/ld0/kernelD0")
    top = names.get("lt/kernel")
    if first is None or top is None:
        return out
    kh, _, cin, cout = first["shape"]
    out["ks"], out["nChannels"], out["nOut0"] = int(kh), int(cin), int(cout)
    out["nClasses"] = int(top["shape"][3])
    layers = 0
    while ("downsampling/ld%d/kernel1" % layers) in names or ("downsampling/ld%d/kernelD%d" % (layers, layers)) in names:
        layers += 1
    out["nLayers"] = layers
    extra = 0
    while ("downsampling/ld0/kernelExtra%d" % extra) in names or ("ld0/kernelExtra%d" % extra) in names:
        extra += 1
    out["nExtraConvs"] = extra
    return out


def blob_from_checkpoint(hp: HParams, ckpt_prefix: str) -> np.ndarray:
    """TF checkpoint -> canonical blob.  Raises KeyError when a variable the graph needs is absent
    (the reference fails the same way: tf NotFoundError on restore when graph and checkpoint disagree)."""
    ck = tfckpt.load_checkpoint(ckpt_prefix)
    tensors = {}
    for name, shape in tensor_specs(hp):
        tf_name = _ckpt_name(hp, name)
        if tf_name not in ck:
            raise KeyError("checkpoint %s has no variable %s (graph/checkpoint mismatch)" % (ckpt_prefix, tf_name))
        tensors[name] = ck[tf_name]
    return blob_from_tensors(hp, tensors)


def random_blob(hp: HParams, seed: int = 20260101) -> np.ndarray:
    """Seeded synthetic weights: N(0, 1/fan_in) filters, non-trivial BN statistics."""
    rng = np.random.default_rng(seed)
    tensors = {}
    for name, shape in tensor_specs(hp):
        if name.endswith(".bn.gamma"):
            t = rng.uniform(0.5, 1.5, shape)
        elif name.endswith(".bn.beta") or name.endswith(".bn.mean"):
            t = rng.normal(0.0, 0.1, shape)
        elif name.endswith(".bn.var"):
            t = rng.uniform(0.5, 1.5, shape)
        else:
            kh, kw, a, b = shape
            fan_in = kh * kw * (b if name.endswith(".wt") else a)  # convT filters are [kh,kw,Cout,Cin]
            t = rng.normal(0.0, np.sqrt(1.0 / fan_in), shape)
        tensors[name] = t.astype(np.float32)
    return blob_from_tensors(hp, tensors)


# hyper-parameters of every model directory the reference ships, for use when the
# model directory itself is unavailable (bench / tests on the GPU box)
KNOWN_HP = {
    "nucleiDAPI": HParams(GRAPH_LEGACY, 128, 1, 3, 16, 2, 5, 1, batchSize=16),
    # the two other models the reference ships WITH weights (legacy graph, 3x3 kernels): models/mousenucleiDAPI's
    # nuclei20x2bin1chan checkpoint (16 features, not the 20 of its hp.data) and models/CytoplasmIncell (2 classes)
    "mousenucleiDAPI": HParams(GRAPH_LEGACY, 256, 1, 3, 16, 3, 3, 1, batchSize=16),
    "CytoplasmIncell": HParams(GRAPH_LEGACY, 128, 1, 2, 24, 2, 3, 1, batchSize=16),
    "nucleiDAPI1-5": HParams(GRAPH_V2, 64, 1, 3, 80, 4, 3, 0, batchSize=32),
    "nucleiDAPILAMIN": HParams(GRAPH_V2, 128, 2, 3, 36, 5, 3, 0, batchSize=24),
    # the metric tile of BASELINE.json (256x256x2): duo widths at imSize 256
    "synthetic-256": HParams(GRAPH_V2, 256, 2, 3, 36, 5, 3, 0, batchSize=8),
}


@dataclass
class ModelArtefacts:
    hp: HParams
    blob: np.ndarray
    mean: float
    std: float


CONVERTED_NAME = "umx_model.npz"   # written by tools/convert_model.py: the "converted once" form of a model directory
HP_ONLY_NAME = "umx_hp.npz"        # hyper-parameters + mean/std of a model whose weight shard is not in the tree


def detect_graph(model_path: str, prefix: str = "model.ckpt") -> int:
    """Which graph builder wrote this model directory, from the checkpoint's variable names: the legacy script names
    its first filter ``downsampling/ld0/kernel1`` (reference UnMicst.py:84), the v2 scripts ``.../kernelD0``
    (UnMicst1-5.py:89)."""
    conv = os.path.join(model_path, CONVERTED_NAME)
    if os.path.exists(conv):
        with np.load(conv) as z:
            return int(z["hp"][0])
    names = tfckpt.read_index(os.path.join(model_path, prefix + ".index"))
    if "downsampling/ld0/kernelD0" in names:
        return GRAPH_V2
    if "downsampling/ld0/kernel1" in names:
        return GRAPH_LEGACY
    raise KeyError("%s: neither a legacy nor a v2 UnMicst checkpoint" % model_path)


def _hp_vector(hp: HParams) -> np.ndarray:
    return np.array([hp.graph, hp.imSize, hp.nChannels, hp.nClasses, hp.nOut0, hp.nLayers, hp.ks, hp.nExtraConvs,
                     hp.featMapsFact, hp.downSampFact, hp.batchSize], dtype=np.int64)


def hparams_from_vector(h) -> HParams:
    h = [int(v) for v in h]
    hp = HParams(graph=h[0], imSize=h[1], nChannels=h[2], nClasses=h[3], nOut0=h[4], nLayers=h[5], ks=h[6],
                 nExtraConvs=h[7], featMapsFact=h[8], downSampFact=h[9], batchSize=h[10])
    hp.validate()
    return hp


def save_converted(art: "ModelArtefacts", model_path: str) -> str:
    """Write the one-time conversion of a model directory (canonical blob + hp + normalisation scalars)."""
    os.makedirs(model_path, exist_ok=True)
    out = os.path.join(model_path, CONVERTED_NAME)
    np.savez(out, blob=np.ascontiguousarray(art.blob, dtype="<f4"), hp=_hp_vector(art.hp),
             mean=np.float64(art.mean), std=np.float64(art.std))
    return out


def load_model_dir(model_path: str, graph: int = None, synthetic_if_missing: bool = False,
                   prefix: str = "model.ckpt") -> ModelArtefacts:
    """Read a model directory: the converted ``umx_model.npz`` if present, else the reference's own artefacts
    (hp.data, datasetMean/StDev pickles, model.ckpt -- reference UnMicst1-5.py:656-681).  ``graph`` None = detect.

    ``synthetic_if_missing``: when the directory has hyper-parameters but no weight shard (the reference ships
    nucleiDAPI1-5 / nucleiDAPILAMIN that way and downloads the shards at image-build time, Dockerfile:5-6), use
    seeded synthetic weights instead of failing -- for plumbing tests only, never silently.

    ``prefix``: checkpoint prefix inside the directory (``saver.restore(sess, modelPath + '/model.ckpt')``, reference
    UnMicst.py:500-503); a directory that holds another run's files (models/mousenucleiDAPI: ``nuclei20x2bin1chan``) is
    converted by naming it, and the hyper-parameters then follow that checkpoint's tensor shapes."""
    conv = os.path.join(model_path, CONVERTED_NAME)
    if os.path.exists(conv):
        with np.load(conv) as z:
            return ModelArtefacts(hparams_from_vector(z["hp"]), np.array(z["blob"]), float(z["mean"]), float(z["std"]))
    hp_only = os.path.join(model_path, HP_ONLY_NAME)
    if os.path.exists(hp_only) and not os.path.exists(os.path.join(model_path, "hp.data")):
        # the shipped stand-ins for nucleiDAPI1-5 / nucleiDAPILAMIN: hyper-parameters + normalisation scalars, no weights
        # (the reference downloads those shards at image-build time, Dockerfile:5-6; drop umx_model.npz next to this file)
        with np.load(hp_only) as z:
            hp, mean, std = hparams_from_vector(z["hp"]), float(z["mean"]), float(z["std"])
        if not synthetic_if_missing:
            raise FileNotFoundError("%s holds hyper-parameters only: convert the model's checkpoint with tools/convert_model.py "
                                    "(-> %s), or set UMX_SYNTHETIC_WEIGHTS=1 to run with seeded synthetic weights"
                                    % (model_path, CONVERTED_NAME))
        return ModelArtefacts(hp, random_blob(hp), mean, std)
    if graph is None:
        graph = detect_graph(model_path, prefix)
    hp_dict = load_pickle(os.path.join(model_path, "hp.data"))
    if prefix != "model.ckpt":
        import warnings
        warnings.warn("%s: converting checkpoint %r, not the directory's model.ckpt -- the layer widths are read from that checkpoint, but "
                      "imSize / batchSize (hp.data) and the normalisation scalars (datasetMean.data / datasetStDev.data) are the "
                      "directory's and may belong to another training run: they cannot be verified against the weights that are loaded"
                      % (model_path, prefix), RuntimeWarning, stacklevel=2)
        hp_dict = hp_dict_from_checkpoint(hp_dict, tfckpt.read_index(os.path.join(model_path, prefix + ".index")))
    hp = hparams_from_dict(hp_dict, graph)
    mean = float(load_pickle(os.path.join(model_path, "datasetMean.data")))
    std = float(load_pickle(os.path.join(model_path, "datasetStDev.data")))
    shard = os.path.join(model_path, prefix + ".data-00000-of-00001")
    if not os.path.exists(shard):
        if not synthetic_if_missing:
            raise FileNotFoundError("%s is missing (the reference downloads it separately, Dockerfile:5-6); "
                                    "set UMX_SYNTHETIC_WEIGHTS=1 to run with seeded synthetic weights" % shard)
        return ModelArtefacts(hp, random_blob(hp), mean, std)
    blob = blob_from_checkpoint(hp, os.path.join(model_path, prefix))
    return ModelArtefacts(hp, blob, mean, std)
