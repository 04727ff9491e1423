#!/usr/bin/env python3
"""Engine set-up time (umx_create: plan search, weight packing, upload) per shipped hyper-parameter set."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unmicst_amd import model, umx
for key, hp in model.KNOWN_HP.items():
    blob = model.random_blob(hp, seed=1)
    for prec in ("f16x3", "f32"):
        t = time.perf_counter()
        eng = umx.Engine(hp, blob, max_batch=256 if hp.imSize >= 256 else 484, precision=prec)
        dt = time.perf_counter() - t
        eng.close()
        print("%-16s %-6s blob %6.1f MB  create %.3f s" % (key, prec, blob.nbytes / 1e6, dt))
