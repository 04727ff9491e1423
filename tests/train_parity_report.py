"""Diagnostic (GPU box): per-tensor gradient error of the HIP training step against the float64 oracle, beside the
error of the same oracle evaluated in float32 (torch CPU) -- the yardstick for what fp32 arithmetic can deliver."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import helpers  # noqa: E402
from oracle import train_oracle as to  # noqa: E402
from unmicst_amd import model, trainer  # noqa: E402
from test_gpu_train import CASES, _batch, _hp, _oracle_opts  # noqa: E402


def main():
    cases = list(CASES)
    if "--full" in sys.argv:           # BASELINE.json configs[4] at full size (about a minute of CPU for the two oracle runs)
        cases.append(("synthetic-256", 8, "duo"))
    for name, B, regime in cases:
        hp = model.KNOWN_HP[name] if name in model.KNOWN_HP else _hp(name)
        opts = trainer.solo_options() if regime == "solo" else trainer.duo_options()
        blob = model.random_blob(hp, seed=21)
        data, labels, weights = _batch(hp, B, 3)
        r64 = to.loss_and_grads(hp, blob, data, labels, weights, _oracle_opts(opts), 0)
        r32 = to.loss_and_grads(hp, blob, data, labels, weights, _oracle_opts(opts), 0, dtype=torch.float32)
        tr = trainer.Trainer(hp, blob, opts, batch=B)
        loss = tr.step(data, labels, weights, apply_update=False)
        g = tr.grads()
        tr.close()
        G, T, W = to.split_blob(hp, g), to.split_blob(hp, r32[3]), to.split_blob(hp, r64[3])
        print("== %s B=%d %s: loss hip %.9g  oracle64 %.9g  oracle32 %.9g" % (name, B, regime, loss[0], r64[0], r32[0]))
        for k in W:
            if not to.trainable(k):
                continue
            s = np.abs(W[k]).max() + 1e-30
            print("   %-14s max|g| %.3e   hip rel err %.2e   torch-fp32 rel err %.2e" %
                  (k, s, np.abs(G[k] - W[k]).max() / s, np.abs(T[k] - W[k]).max() / s))


if __name__ == "__main__":
    main()
