"""The training-step oracle (oracle/train_oracle.py) checked on the CPU: its inference-mode forward against the C
restatement that the reference's goldens pin, its gradients against central finite differences, the dropout stream,
and the optimiser arithmetic against a hand-written scalar restatement."""
import numpy as np
import pytest

from oracle import oracle as orc
from oracle import train_oracle as to
from unmicst_amd import model

from helpers import small_hps


def _batch(hp, B, seed):
    rng = np.random.default_rng(seed)
    data = rng.normal(0, 1, (B, hp.imSize, hp.imSize, hp.nChannels))
    cls = rng.integers(0, hp.nClasses, (B, hp.imSize, hp.imSize))
    labels = np.eye(hp.nClasses)[cls]
    weights = rng.uniform(0.5, 3.0, labels.shape)
    return data, labels, weights


@pytest.mark.parametrize("name", ["v2_solo_like", "v2_duo_like", "v2_deep"])
def test_inference_mode_matches_c_oracle(name):
    hp = small_hps()[name]
    blob = model.random_blob(hp, seed=5)
    data, _, _ = _batch(hp, 3, 1)
    want = orc.forward(hp, blob, data.astype(np.float32))
    got = to.inference_probs(hp, blob, data.astype(np.float32))
    assert np.abs(got - want).max() < 2e-6


@pytest.mark.parametrize("key", ["nucleiDAPI1-5", "nucleiDAPILAMIN"])
def test_inference_mode_matches_c_oracle_at_the_shipped_widths(key):
    """The two independent restatements of the v2 graph (oracle/unet_oracle.c and the torch one here) at the FULL width of
    the shipped solo (80..1280 channels) and duo (36..1152) hyper-parameters, seeded weights with non-trivial BN statistics:
    the reference pins neither (no weights, no outputs in its tree -- SURVEY.md section 8c), so the two must at least pin
    each other."""
    hp = model.KNOWN_HP[key]
    blob = model.random_blob(hp, seed=20260101)
    data = np.random.default_rng(2).normal(0, 1, (1, hp.imSize, hp.imSize, hp.nChannels)).astype(np.float32)
    want = orc.forward(hp, blob, data)
    got = to.inference_probs(hp, blob, data)
    assert np.abs(got - want).max() < 5e-6


def test_specs_agree_with_product_blob_order():
    for name in ("v2_solo_like", "v2_duo_like", "v2_wide"):
        hp = small_hps()[name]
        assert to.tensor_specs(hp) == [(n, tuple(s)) for n, s in model.tensor_specs(hp)]


def test_dropout_stream_statistics_and_determinism():
    m1 = to.dropout_mask(7, 3, to.LAYER_BOTTOM, (4, 16, 16, 8), 0.35)
    m2 = to.dropout_mask(7, 3, to.LAYER_BOTTOM, (4, 16, 16, 8), 0.35)
    m3 = to.dropout_mask(7, 4, to.LAYER_BOTTOM, (4, 16, 16, 8), 0.35)
    assert np.array_equal(m1, m2) and not np.array_equal(m1, m3)
    kept = (m1 > 0).mean()
    assert abs(kept - 0.65) < 0.02
    assert np.allclose(m1[m1 > 0], 1 / (1 - np.float32(0.35)), rtol=1e-7)
    assert np.array_equal(to.dropout_mask(7, 3, 0, (2, 2, 2, 2), 0.0), np.ones((2, 2, 2, 2)))


@pytest.mark.parametrize("opts", [to.solo_options(), to.duo_options()], ids=["solo", "duo"])
def test_gradients_match_finite_differences(opts):
    hp = model.HParams(model.GRAPH_V2, 16, 2, 3, 4, 2, 3, 0)
    blob = model.random_blob(hp, seed=3).astype(np.float64)
    data, labels, weights = _batch(hp, 2, 9)
    loss, _, _, g, _, _ = to.loss_and_grads(hp, blob, data, labels, weights, opts, step=2)
    rng = np.random.default_rng(0)
    specs = to.tensor_specs(hp)
    pos = 0
    for name, shape in specs:
        n = int(np.prod(shape))
        if to.trainable(name):
            for j in rng.choice(n, size=min(3, n), replace=False):
                k = pos + int(j)
                h = 1e-6 * max(1.0, abs(blob[k]))
                bp, bm = blob.copy(), blob.copy()
                bp[k] += h
                bm[k] -= h
                lp = to.loss_and_grads(hp, bp, data, labels, weights, opts, step=2)[0]
                lm = to.loss_and_grads(hp, bm, data, labels, weights, opts, step=2)[0]
                fd = (lp - lm) / (2 * h)
                assert abs(fd - g[k]) <= 2e-5 * max(1.0, abs(g[k])) + 1e-7, (name, j, fd, g[k])
        else:
            assert not g[pos:pos + n].any()
        pos += n


def test_adam_and_moving_statistics_arithmetic():
    hp = model.HParams(model.GRAPH_V2, 16, 1, 3, 4, 2, 3, 0)
    o = to.solo_options(decay_steps=2)
    st = to.TrainState(model.random_blob(hp, seed=11))
    b0 = st.blob.copy()
    m = np.zeros_like(b0)
    v = np.zeros_like(b0)
    w = b0.copy()
    losses = []
    for step in range(3):
        data, labels, weights = _batch(hp, 2, 100 + step)
        prev = st.blob.copy()
        losses.append(to.train_step(hp, st, data, labels, weights, o))
        g = st.last["grads"]
        lr = 5e-5 * 0.98 ** (step // 2)
        assert st.last["lr"] == pytest.approx(lr)
        t = step + 1
        m = 0.9 * m + 0.1 * g
        v = 0.999 * v + 0.001 * g * g
        upd = lr * np.sqrt(1 - 0.999 ** t) / (1 - 0.9 ** t) * m / (np.sqrt(v) + 1e-8)
        T_prev, T_now = to.split_blob(hp, prev), to.split_blob(hp, st.blob)
        U = to.split_blob(hp, upd)
        for name in T_now:
            if to.trainable(name):
                assert np.allclose(T_now[name], T_prev[name] - U[name], rtol=0, atol=1e-15)
        mean, var, n = st.last["stats"]["ld0"]
        assert np.allclose(T_now["ld0.bn.mean"], 0.99 * T_prev["ld0.bn.mean"] + 0.01 * mean)
        assert np.allclose(T_now["ld0.bn.var"], 0.99 * T_prev["ld0.bn.var"] + 0.01 * var * n / (n - 1))
    assert st.step == 3 and all(np.isfinite(losses))


def test_momentum_step():
    hp = model.HParams(model.GRAPH_V2, 16, 1, 3, 4, 1, 3, 0)
    o = to.TrainOptions(optimizer="momentum", lr0=0.01, decay_steps=1000, decay_rate=0.95, reg_kind=0, drop_bottom=0.0)
    st = to.TrainState(model.random_blob(hp, seed=2))
    data, labels, weights = _batch(hp, 2, 5)
    w0 = st.blob.copy()
    l0 = to.train_step(hp, st, data, labels, weights, o)
    g0 = st.last["grads"]
    l1 = to.train_step(hp, st, data, labels, weights, o)
    g1 = st.last["grads"]
    T0, T2 = to.split_blob(hp, w0), to.split_blob(hp, st.blob)
    G0, G1 = to.split_blob(hp, g0), to.split_blob(hp, g1)
    for name in T2:
        if to.trainable(name):
            assert np.allclose(T2[name], T0[name] - 0.01 * G0[name] - 0.01 * (0.9 * G0[name] + G1[name]), atol=1e-14)
    assert l1 < l0   # the same batch twice with plain momentum SGD: the loss goes down
