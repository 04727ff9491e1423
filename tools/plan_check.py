#!/usr/bin/env python3
"""Will this model run on the split-precision (fast) kernels, or fall back to the exact-fp32 engine?  Host only, no GPU needed.

usage: plan_check.py <model dir | known model name> [...]
    A model directory holds umx_model.npz / umx_hp.npz (tools/convert_model.py) or the reference's hp.data; a known name is one of
    unmicst_amd.model.KNOWN_HP (nucleiDAPI, nucleiDAPI1-5, nucleiDAPILAMIN, synthetic-256, ...).
Prints, per model, "split precision" or the first layer the planner refuses and why (umx_plan_check, include/umx.h).
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from unmicst_amd import model, umx  # noqa: E402


def main(argv):
    if not argv:
        print(__doc__)
        return 2
    rc = 0
    for a in argv:
        if a in model.KNOWN_HP:
            hp = model.KNOWN_HP[a]
        else:
            hp = model.load_model_dir(a, synthetic_if_missing=True).hp
        why = umx.plan_check(hp)
        print("%-28s %s" % (a, "split precision (f16x3)" if not why else "exact fp32 (about 4.6 x slower): " + why))
        rc = rc or (1 if why else 0)
    return rc


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
