#!/usr/bin/env python3
"""Parity of the fp6 cross-term form (UMX_PREC_F16X3_F6, conv_f16x3's F6 kernels) on the GPU: max |p - p_oracle| per tile over the
softmax outputs next to the 3-product engine, on the graphs whose layers it touches (9-tile kernels at <= 1/4 resolution).  GPU box."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import helpers  # noqa: E402
from oracle import oracle  # noqa: E402
from unmicst_amd import model, umx  # noqa: E402


def main():
    cases = {"v2 64 px, widths 72..288": model.HParams(model.GRAPH_V2, 64, 2, 3, 72, 2, 3, 0),
             "duo hp (128x128x2, widths 36..1152)": model.KNOWN_HP["nucleiDAPILAMIN"],
             "synthetic-256": model.KNOWN_HP["synthetic-256"]}
    print("%-40s %12s %12s %12s" % ("case (max |p_gpu - p_oracle|)", "f16x3", "f16f6", "f16f6-f16x3"))
    for name, hp in cases.items():
        for seed in (11, 12):
            blob = model.random_blob(hp, seed=seed)
            n = 1 if hp.imSize >= 256 else 2 if hp.imSize >= 128 else 5
            x = np.random.default_rng(5 + seed).normal(size=(n, hp.imSize, hp.imSize, hp.nChannels)).astype(np.float32)
            ref = oracle.forward(hp, blob, x)
            got = {}
            for prec in ("f16x3", "f16f6"):
                with umx.Engine(hp, blob, max_batch=4, precision=prec) as eng:
                    got[prec] = eng.forward_tiles(x)
                    assert eng.precision == prec, (eng.precision, prec)   # (f16f6 is reported only where a layer takes the form)
            print("%-40s %12.3g %12.3g %12.3g" % ("%s seed %d" % (name, seed), np.abs(got["f16x3"] - ref).max(), np.abs(got["f16f6"] - ref).max(),
                                                  np.abs(got["f16f6"] - got["f16x3"]).max()), flush=True)


if __name__ == "__main__":
    main()
