#!/usr/bin/env python3
"""Print the measured parity of the HIP path against the oracle for both conv precisions (GPU box; table for DESIGN.md).
The oracle is the checker here, exactly as in tests/test_gpu_parity.py."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))   # repo root
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))  # (this file lives in tests/: the oracle may only be used from there)

import helpers  # noqa: E402
from oracle import oracle, pi2d_oracle  # noqa: E402
from unmicst_amd import model, umx  # noqa: E402


def main():
    cases = dict(helpers.small_hps())
    cases["duo hp (128x128x2, widths 36..1152)"] = model.KNOWN_HP["nucleiDAPILAMIN"]
    print("%-40s %12s %12s" % ("case (max |p_gpu - p_oracle| over softmax outputs)", "f32", "f16x3"))
    for name, hp in cases.items():
        blob = model.random_blob(hp, seed=11)
        n = 2 if hp.imSize >= 128 else 5
        x = np.random.default_rng(5).normal(size=(n, hp.imSize, hp.imSize, hp.nChannels)).astype(np.float32)
        ref = oracle.forward(hp, blob, x)
        errs = []
        for prec in ("f32", "f16x3"):
            with umx.Engine(hp, blob, max_batch=4, precision=prec) as eng:
                errs.append(float(np.abs(eng.forward_tiles(x) - ref).max()))
        print("%-40s %12.3g %12.3g" % (name, errs[0], errs[1]))
    hp, blob, mean, std = helpers.load_nuclei_dapi()
    raw, g_cont, g_raw, g_nuc = helpers.load_sample_105()
    I = helpers.legacy_preprocess(raw)
    pi = pi2d_oracle.PI2DOracle(I, hp.imSize, hp.margin, "accumulate")
    x = pi2d_oracle.normalised_batch(pi, 20, 16, 1, mean, std, False)
    ref = oracle.forward(hp, blob, x)
    for prec in ("f32", "f16x3"):
        with umx.Engine(hp, blob, max_batch=32, precision=prec) as eng:
            e = float(np.abs(eng.forward_tiles(x) - ref).max())
            planes = eng.infer_image(I, mean, std)
        stats = []
        for k, gold in ((1, g_cont), (2, g_nuc)):
            pm = np.uint8(255 * planes[k].astype(np.float64))
            d = np.abs(pm.astype(int) - gold.astype(int))
            stats.append("class %d: max %d LSB, %.2f%% exact" % (k, d.max(), 100 * (d == 0).mean()))
        print("nucleiDAPI real weights, 105.tif  %-6s tile err %.3g | vs reference goldens: %s" % (prec, e, "; ".join(stats)))


if __name__ == "__main__":
    main()
