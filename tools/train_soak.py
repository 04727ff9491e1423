#!/usr/bin/env python3
"""Soak run of the training step: N steps on a few fixed random batches (the model must overfit them: the loss falls),
watching for non-finite values and range errors.  usage: train_soak.py [steps] [regime]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from unmicst_amd import model, trainer

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
regime = sys.argv[2] if len(sys.argv) > 2 else "duo"
hp = model.KNOWN_HP["synthetic-256"]
opts = (trainer.duo_options if regime == "duo" else trainer.solo_options)(lr0=1e-3)   # faster than the scripts' 5e-5: a soak, not a recipe
rng = np.random.default_rng(0)
B, nb = 8, 2
data = rng.normal(size=(nb, B, hp.imSize, hp.imSize, hp.nChannels)).astype(np.float32)
cls = rng.integers(0, hp.nClasses, (nb, B, hp.imSize // 16, hp.imSize // 16)).repeat(16, 2).repeat(16, 3)   # blocky labels: learnable
labels = np.eye(hp.nClasses, dtype=np.float32)[cls]
data[..., 0] += cls                                                                                         # ... from the image
weights = np.ones_like(labels)
tr = trainer.Trainer(hp, model.random_blob(hp, seed=3), opts, batch=B)
t0 = time.perf_counter()
hist = []
for s in range(steps):
    loss, dt, reg = tr.step(data[s % nb], labels[s % nb], weights[s % nb])
    hist.append(dt)
    if s % 25 == 0 or s == steps - 1:
        print("step %4d  loss %.5f  data term %.5f  reg %.5f" % (s, loss, dt, reg), flush=True)
    assert np.isfinite(loss), "non-finite loss at step %d" % s
blob = tr.blob()
assert np.isfinite(blob).all()
print("%.1f steps/s; data term %.4f -> %.4f" % (steps / (time.perf_counter() - t0), hist[0], hist[-1]))
assert hist[-1] < 0.5 * hist[0], "the loss did not fall"
print("soak ok")
