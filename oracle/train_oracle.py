"""ORACLE -- test infrastructure only (never imported by the product path).

CPU restatement (torch autograd, float64 by default) of ONE optimisation step of the reference's v2 training graph:

* graph in training mode: reference UnMicst1-5.py:83-237 (solo) / UnMicst2.py:80-235 (duo) -- batch-statistics BN
  (``tf.layers.batch_normalization(training=True)``, eps 1e-3 from the checkpoints' .meta), LeakyReLU(0.2),
  ks x ks shortcut added to the main conv, 2x2 max-pool, stride-2 SAME transposed conv, concat [skip, up], 1x1 top conv +
  BN + softmax, dropout where the script applies it (solo: bottom 0.35, UnMicst1-5.py:139; duo: down 0.05*i, bottom
  0.3, up 0.25-0.05*i, UnMicst2.py:114,137,203);
* loss: ``mean_{b,y,x}( -sum_k w*y*log(clip(p, 1e-7, 1-1e-7)) ) + regularisation`` (UnMicst1-5.py:362-367; duo takes
  log(p) without the clip, UnMicst2.py:363-366); regularisers only on the variables created with ``regularizer=``
  (shortcut, bottom, up and top kernels): l1(8e-5) solo (UnMicst1-5.py:84,125,160,213), l2(0.01 / 0.005) duo
  (UnMicst2.py:82,123,158,211);
* optimiser: tf.train.AdamOptimizer(lr0 * rate^floor(step/decay_steps)) (UnMicst1-5.py:355-373, UnMicst2.py:357-371),
  TF1 update rule (lr_t = lr*sqrt(1-b2^t)/(1-b1^t); w -= lr_t*m/(sqrt(v)+1e-8)); MomentumOptimizer(lr, 0.9) is the
  legacy script's (UnMicst.py:270-279) and is restated for completeness;
* BN moving statistics: momentum 0.99; the fused BN kernel feeds the *unbiased* batch variance to the moving average.

PARITY UNPINNED: TensorFlow is not installable here and the reference tree holds no training outputs, so nothing pins
these numerics except (i) the inference-mode forward of this file agreeing with oracle/unet_oracle.c (which the
reference's 105.tif goldens pin for the shared primitives) and (ii) finite-difference checks of the gradients
(tests/test_train_oracle.py).  Dropout masks cannot match TF's Philox stream; the mask here is a documented
counter-based hash (``dropout_mask``) that the HIP kernels restate bit for bit.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Dict, List, Tuple

import numpy as np
import torch
import torch.nn.functional as F

BN_EPS = 1e-3
LEAK = 0.2

LAYER_DOWN, LAYER_BOTTOM, LAYER_UP = 0, 16, 32   # dropout stream ids: down i -> i, bottom -> 16, up idx -> 32+idx


@dataclass
class TrainOptions:
    optimizer: str = "adam"          # "adam" | "momentum"
    lr0: float = 5e-5
    decay_steps: int = 5000
    decay_rate: float = 0.98
    momentum: float = 0.9
    beta1: float = 0.9
    beta2: float = 0.999
    adam_eps: float = 1e-8
    reg_kind: int = 1                # 0 none, 1 L1, 2 L2
    reg_down: float = 8e-5           # shortcut kernels
    reg_bottom: float = 8e-5
    reg_up: float = 8e-5             # transposed-conv and conv kernels of the up layers
    reg_top: float = 8e-5
    clip_eps: float = 1e-7           # 0: no clip (duo)
    drop_down_step: float = 0.0      # down layer i: drop_down_step * i
    drop_bottom: float = 0.35
    drop_up0: float = 0.0            # up layer idx: drop_up0 - drop_up_step * idx
    drop_up_step: float = 0.0
    bn_momentum: float = 0.99
    seed: int = 1234


def solo_options(**kw) -> TrainOptions:
    return TrainOptions(**kw)


def duo_options(**kw) -> TrainOptions:
    base = dict(lr0=6e-5, decay_steps=4000, decay_rate=0.99, reg_kind=2, reg_down=0.01, reg_bottom=0.01, reg_up=0.005,
                reg_top=0.005, clip_eps=0.0, drop_down_step=0.05, drop_bottom=0.3, drop_up0=0.25, drop_up_step=0.05)
    base.update(kw)
    return TrainOptions(**base)


def tensor_specs(hp) -> List[Tuple[str, Tuple[int, ...]]]:
    """Blob order of the v2 graph (same order as the product's weight blob; restated so the oracle stands alone)."""
    n = [hp.nChannels, hp.nOut0]
    for _ in range(hp.nLayers):
        n.append(n[-1] * hp.featMapsFact)
    ks = hp.ks
    specs = []

    def bn(prefix, c):
        for t in ("gamma", "beta", "mean", "var"):
            specs.append(("%s.bn.%s" % (prefix, t), (c,)))

    for i in range(hp.nLayers):
        specs.append(("ld%d.w1" % i, (ks, ks, n[i], n[i + 1])))
        specs.append(("ld%d.wshort" % i, (ks, ks, n[i], n[i + 1])))
        bn("ld%d" % i, n[i + 1])
    specs.append(("lb.w", (ks, ks, n[hp.nLayers], n[hp.nLayers + 1])))
    bn("lb", n[hp.nLayers + 1])
    for idx in range(hp.nLayers - 1, -1, -1):
        specs.append(("lu%d.wt" % idx, (ks, ks, n[idx + 1], n[idx + 2])))
        specs.append(("lu%d.w2" % idx, (ks, ks, n[idx] + n[idx + 1], n[idx + 1])))
        bn("lu%d" % idx, n[idx + 1])
    specs.append(("lt.w", (1, 1, n[1], hp.nClasses)))
    bn("lt", hp.nClasses)
    return specs


def split_blob(hp, blob: np.ndarray) -> Dict[str, np.ndarray]:
    if hp.nExtraConvs != 0:
        raise ValueError("training restatement covers nExtraConvs == 0 (every shipped v2 model)")
    out, pos = {}, 0
    for name, shape in tensor_specs(hp):
        k = int(np.prod(shape))
        out[name] = np.asarray(blob[pos:pos + k]).reshape(shape)
        pos += k
    if pos != np.asarray(blob).size:
        raise ValueError("blob has %d floats, graph needs %d" % (np.asarray(blob).size, pos))
    return out


def join_blob(hp, tensors: Dict[str, np.ndarray]) -> np.ndarray:
    return np.concatenate([np.asarray(tensors[name], dtype=np.float64).ravel() for name, _ in tensor_specs(hp)])


def trainable(name: str) -> bool:
    return not (name.endswith(".bn.mean") or name.endswith(".bn.var"))


def reg_coef(name: str, o: TrainOptions) -> float:
    if o.reg_kind == 0:
        return 0.0
    if name.endswith(".wshort"):
        return o.reg_down
    if name == "lb.w":
        return o.reg_bottom
    if name.endswith(".wt") or name.endswith(".w2"):
        return o.reg_up
    if name == "lt.w":
        return o.reg_top
    return 0.0   # kernelD%d is a bare tf.Variable (UnMicst1-5.py:85-87): no regulariser; BN variables: none


# ------------------------------------------------------------------------------------------------ dropout stream
_M64 = (1 << 64) - 1


def _mix(x: np.ndarray) -> np.ndarray:
    x = np.asarray(x, dtype=np.uint64)
    with np.errstate(over="ignore"):
        x = (x ^ (x >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        x = (x ^ (x >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        x = x ^ (x >> np.uint64(31))
    return x


def dropout_mask(seed: int, step: int, layer_id: int, shape_nhwc, rate: float) -> np.ndarray:
    """Multiplier of every NHWC element: 0 (dropped) or 1/(1-rate) -- tf.layers.dropout semantics with a counter-based
    stream: u = top 24 bits of mix(mix(seed + GOLDEN*(64*step + layer_id + 1)) ^ flat_index); keep iff u >= rate."""
    n = int(np.prod(shape_nhwc))
    if rate <= 0.0:
        return np.ones(shape_nhwc, dtype=np.float64)
    key = _mix(np.uint64((seed + 0x9E3779B97F4A7C15 * (64 * step + layer_id + 1)) & _M64))
    h = _mix(np.arange(n, dtype=np.uint64) ^ key)
    u = (h >> np.uint64(40)).astype(np.float64) * (1.0 / (1 << 24))
    keep = u >= np.float32(rate).astype(np.float64)
    scale = np.float32(1.0) / (np.float32(1.0) - np.float32(rate))    # the kernels form the scale in fp32
    return (keep * np.float64(scale)).reshape(shape_nhwc)


# ------------------------------------------------------------------------------------------------ graph
def _conv_same(x, w_tf):
    """x NCHW, w_tf [kh,kw,Cin,Cout] (tf.nn.conv2d, stride 1, SAME, odd kernel)."""
    w = w_tf.permute(3, 2, 0, 1)
    return F.conv2d(x, w, padding=(w_tf.shape[0] // 2, w_tf.shape[1] // 2))


def _conv_transpose_s2(x, wt_tf):
    """tf.nn.conv2d_transpose(stride 2, SAME, output 2H): wt_tf [kh,kw,Cout,Cin]; the gradient of a stride-2 SAME conv
    whose padding is (ks-2)//2 before and the rest after (crop rule of SURVEY.md section 2.1)."""
    kh, kw = wt_tf.shape[0], wt_tf.shape[1]
    full = F.conv_transpose2d(x, wt_tf.permute(3, 2, 0, 1), stride=2)   # [2H + kh - 2]
    pbh, pbw = max(kh - 2, 0) // 2, max(kw - 2, 0) // 2
    H2, W2 = 2 * x.shape[2], 2 * x.shape[3]
    return full[:, :, pbh:pbh + H2, pbw:pbw + W2]


def _bn(x, P, prefix, training, stats):
    g, b = P[prefix + ".bn.gamma"], P[prefix + ".bn.beta"]
    if training:
        mean = x.mean(dim=(0, 2, 3))
        var = x.var(dim=(0, 2, 3), unbiased=False)
        n = x.numel() // x.shape[1]
        stats[prefix] = (mean.detach(), var.detach(), n)
    else:
        mean, var = P[prefix + ".bn.mean"], P[prefix + ".bn.var"]
    xh = (x - mean[None, :, None, None]) * torch.rsqrt(var[None, :, None, None] + BN_EPS)
    return xh * g[None, :, None, None] + b[None, :, None, None]


def _drop(x, rate, o, step, layer_id, training):
    if not training or rate <= 0.0:
        return x
    B, C, H, W = x.shape
    m = dropout_mask(o.seed, step, layer_id, (B, H, W, C), rate)
    return x * torch.from_numpy(m).permute(0, 3, 1, 2).to(x.dtype)


def _leaky(a, site, decisions, trace):
    """LeakyReLU(0.2) -- or, with `decisions[site]` (a tensor of slopes 1 / 0.2 shaped like `a`), the same function with every
    branch TAKEN AS GIVEN: a * slope.  Where the given branch is the one this run would take the two are identical; where an
    activation is within rounding distance of 0 the value moves by < |a| while the gradient is the given branch's."""
    if trace is not None:
        trace[site] = a.detach()
    if decisions is not None and site in decisions:
        return a * decisions[site]
    return F.leaky_relu(a, LEAK)


def _pool(a, site, decisions, trace):
    """max_pool2d(2) -- or, with `decisions[site]` (int64 [B,C,H/2,W/2], window element 2*dy + dx), the given element of every window."""
    if trace is not None:
        trace[site] = a.detach()
    if decisions is not None and site in decisions:
        B, C, H, W = a.shape
        win = a.reshape(B, C, H // 2, 2, W // 2, 2).permute(0, 1, 2, 4, 3, 5).reshape(B, C, H // 2, W // 2, 4)
        return torch.gather(win, 4, decisions[site].unsqueeze(-1)).squeeze(-1)
    return F.max_pool2d(a, 2)


def forward(hp, P: Dict[str, torch.Tensor], data_nhwc: torch.Tensor, o: TrainOptions, step: int, training: bool,
            decisions=None, trace=None):
    """-> (softmax probabilities NHWC, BN batch statistics per layer).

    decisions (test aid, tests/test_gpu_train.py): {"ld<i>" | "lb" | "lu<i>" | "us<i>": slopes, "pool<i>": window elements} -- the
    LeakyReLU branches and max-pool choices of ANOTHER evaluation of the same step (the HIP kernels'), so that two implementations
    are compared on the same smooth piece of the loss; trace: dict that receives the tensor in front of every such decision."""
    L = hp.nLayers
    stats: Dict[str, tuple] = {}
    x = data_nhwc.permute(0, 3, 1, 2)
    ds = [x]
    for i in range(L):
        z = _conv_same(ds[i], P["ld%d.w1" % i]) + _conv_same(ds[i], P["ld%d.wshort" % i])
        a = _leaky(_bn(z, P, "ld%d" % i, training, stats), "ld%d" % i, decisions, trace)
        a = _drop(a, o.drop_down_step * i, o, step, LAYER_DOWN + i, training)
        ds.append(_pool(a, "pool%d" % i, decisions, trace))
    b = _leaky(_bn(_conv_same(ds[L], P["lb.w"]), P, "lb", training, stats), "lb", decisions, trace)
    cur = _drop(b, o.drop_bottom, o, step, LAYER_BOTTOM, training)
    for idx in range(L - 1, -1, -1):
        us = _leaky(_conv_transpose_s2(cur, P["lu%d.wt" % idx]), "us%d" % idx, decisions, trace)
        cc = torch.cat([ds[idx], us], dim=1)
        cv = _leaky(_bn(_conv_same(cc, P["lu%d.w2" % idx]), P, "lu%d" % idx, training, stats), "lu%d" % idx, decisions, trace)
        cur = _drop(cv, o.drop_up0 - o.drop_up_step * idx, o, step, LAYER_UP + idx, training)
    t = _bn(_conv_same(cur, P["lt.w"]), P, "lt", training, stats)
    return torch.softmax(t, dim=1).permute(0, 2, 3, 1), stats


def loss_of(hp, P, probs_nhwc, labels, weights, o: TrainOptions):
    p = probs_nhwc
    if o.clip_eps > 0:
        p = torch.clamp(p, o.clip_eps, 1.0 - o.clip_eps)
    data_term = (-(weights * labels * torch.log(p)).sum(dim=3)).mean()
    reg = torch.zeros((), dtype=p.dtype)
    for name, t in P.items():
        c = reg_coef(name, o)
        if c > 0:
            reg = reg + c * (t.abs().sum() if o.reg_kind == 1 else (t * t).sum())
    return data_term + reg, data_term, reg


def learning_rate(o: TrainOptions, global_step: int) -> float:
    return o.lr0 * o.decay_rate ** (global_step // o.decay_steps)


@dataclass
class TrainState:
    blob: np.ndarray                        # float64 parameters (+ BN moving statistics), blob order
    m: np.ndarray = None
    v: np.ndarray = None
    step: int = 0
    last: dict = field(default_factory=dict)

    def __post_init__(self):
        self.blob = np.asarray(self.blob, dtype=np.float64).copy()
        if self.m is None:
            self.m = np.zeros_like(self.blob)
        if self.v is None:
            self.v = np.zeros_like(self.blob)


def loss_and_grads(hp, blob, data, labels, weights, o: TrainOptions, step: int, dtype=torch.float64, decisions=None, trace=None):
    """-> (loss, data_term, reg, grads as a blob-shaped float64 vector (0 for the moving statistics), probs, stats).
    decisions / trace: see forward()."""
    T = split_blob(hp, np.asarray(blob, dtype=np.float64))
    P = {k: torch.tensor(v, dtype=dtype, requires_grad=trainable(k)) for k, v in T.items()}
    d = torch.tensor(np.asarray(data), dtype=dtype)
    y = torch.tensor(np.asarray(labels), dtype=dtype)
    w = torch.tensor(np.asarray(weights), dtype=dtype)
    probs, stats = forward(hp, P, d, o, step, training=True, decisions=decisions, trace=trace)
    loss, data_term, reg = loss_of(hp, P, probs, y, w, o)
    loss.backward()
    grads = {k: (P[k].grad.numpy() if trainable(k) and P[k].grad is not None else np.zeros(T[k].shape)) for k in T}
    return (loss.item(), data_term.item(), float(reg.detach()), join_blob(hp, grads), probs.detach().numpy(),
            {k: (m.numpy(), v.numpy(), n) for k, (m, v, n) in stats.items()})


def train_step(hp, st: TrainState, data, labels, weights, o: TrainOptions, dtype=torch.float64) -> float:
    """One optimisation step in place (parameters, optimiser slots, BN moving statistics, step counter)."""
    loss, data_term, reg, g, probs, stats = loss_and_grads(hp, st.blob, data, labels, weights, o, st.step, dtype)
    lr = learning_rate(o, st.step)
    names = tensor_specs(hp)
    mask = np.concatenate([np.full(int(np.prod(s)), 1.0 if trainable(n) else 0.0) for n, s in names])
    t = st.step + 1
    if o.optimizer == "adam":
        lr_t = lr * np.sqrt(1.0 - o.beta2 ** t) / (1.0 - o.beta1 ** t)
        st.m = o.beta1 * st.m + (1.0 - o.beta1) * g
        st.v = o.beta2 * st.v + (1.0 - o.beta2) * g * g
        st.blob = st.blob - mask * lr_t * st.m / (np.sqrt(st.v) + o.adam_eps)
    elif o.optimizer == "momentum":
        st.m = o.momentum * st.m + g            # tf.train.MomentumOptimizer: accum = mom*accum + g; w -= lr*accum
        st.blob = st.blob - mask * lr * st.m
    else:
        raise ValueError(o.optimizer)
    T = split_blob(hp, st.blob)
    for prefix, (mean, var, n) in stats.items():
        mm, mv = T[prefix + ".bn.mean"], T[prefix + ".bn.var"]
        mm[...] = mm * o.bn_momentum + mean * (1.0 - o.bn_momentum)
        mv[...] = mv * o.bn_momentum + var * (n / max(n - 1.0, 1.0)) * (1.0 - o.bn_momentum)
    st.step = t
    st.last = {"loss": loss, "data_term": data_term, "reg": reg, "grads": g, "probs": probs, "lr": lr, "stats": stats}
    return loss


def inference_probs(hp, blob, data, dtype=torch.float64) -> np.ndarray:
    """Inference-mode forward (moving statistics, no dropout) -- used to tie this restatement to unet_oracle.c."""
    T = split_blob(hp, np.asarray(blob, dtype=np.float64))
    P = {k: torch.tensor(v, dtype=dtype) for k, v in T.items()}
    with torch.no_grad():
        probs, _ = forward(hp, P, torch.tensor(np.asarray(data), dtype=dtype), TrainOptions(), 0, training=False)
    return probs.numpy()
