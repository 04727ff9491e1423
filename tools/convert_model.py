#!/usr/bin/env python3
"""One-time conversion of a reference-format model directory (hp.data, datasetMean/StDev pickles, TensorFlow
model.ckpt) into ``umx_model.npz`` (canonical float32 weight blob + hyper-parameters + normalisation scalars).
No TensorFlow needed: the checkpoint is read by unmicst_amd/tfckpt.py.

usage: convert_model.py <model dir> [<output dir>]      (output dir defaults to the model dir)
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from unmicst_amd import model  # noqa: E402


def main():
    if len(sys.argv) < 2:
        sys.exit(__doc__)
    src = sys.argv[1]
    dst = sys.argv[2] if len(sys.argv) > 2 else src
    art = model.load_model_dir(src)
    out = model.save_converted(art, dst)
    print("%s: graph %s, %d floats (%.1f MB), mean %.6g std %.6g -> %s" % (
        src, "v2" if art.hp.graph else "legacy", art.blob.size, art.blob.nbytes / 1e6, art.mean, art.std, out))


if __name__ == "__main__":
    main()
