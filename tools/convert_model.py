#!/usr/bin/env python3
"""One-time conversion of a reference-format model directory (hp.data, datasetMean/StDev pickles, TensorFlow
model.ckpt) into ``umx_model.npz`` (canonical float32 weight blob + hyper-parameters + normalisation scalars).
No TensorFlow needed: the checkpoint is read by unmicst_amd/tfckpt.py.

usage: convert_model.py <model dir> [<output dir>] [--prefix NAME]
    output dir defaults to the model dir; --prefix names the checkpoint inside the directory (default ``model.ckpt``,
    what the reference restores: UnMicst.py:500-503).  models/mousenucleiDAPI holds its weights under ``nuclei20x2bin1chan``
    (its ``model.ckpt`` shard is missing from the reference tree): ``--prefix nuclei20x2bin1chan`` converts that one, and
    the hyper-parameters then follow the checkpoint's tensor shapes where they differ from hp.data.
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from unmicst_amd import model  # noqa: E402


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("model_dir")
    ap.add_argument("output_dir", nargs="?")
    ap.add_argument("--prefix", default="model.ckpt")
    a = ap.parse_args(argv)
    art = model.load_model_dir(a.model_dir, prefix=a.prefix)
    out = model.save_converted(art, a.output_dir or a.model_dir)
    print("%s/%s: graph %s, %dx%dx%d tile, %d classes, ks %d, nOut0 %d, %d layers, %d floats (%.1f MB), mean %.6g std %.6g -> %s" % (
        a.model_dir, a.prefix, "v2" if art.hp.graph else "legacy", art.hp.imSize, art.hp.imSize, art.hp.nChannels,
        art.hp.nClasses, art.hp.ks, art.hp.nOut0, art.hp.nLayers, art.blob.size, art.blob.nbytes / 1e6, art.mean, art.std, out))
    return out


if __name__ == "__main__":
    main()
