#!/usr/bin/env python3
"""End-to-end wall time of the reference-compatible CLI on a synthetic slide (GPU box): UnMicst.py (legacy tool, the shipped
nucleiDAPI weights) on an N x N uint16 TIFF, with the per-stage breakdown the driver prints under UMX_CLI_TIMING=1
(interpreter start + imports, engine set-up, TIFF read, engine call, page writes, clean-up).
usage: cli_walltime.py [N=16384] [other-tree ...]   -- every further argument is another checkout of the repo (with its own built
libumx.so) whose UnMicst.py is timed on the same file, same box: how a round's change is compared with the round before."""
import json
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import helpers  # noqa: E402
from unmicst_amd import model, tiffio  # noqa: E402


def one(tree, path, out, env_extra, label, n, script="UnMicst.py", what="legacy nucleiDAPI"):
    e = dict(os.environ, UMX_CLI_TIMING="1", **env_extra)
    e["UMX_CLI_T0"] = repr(time.time())
    t = time.perf_counter()
    r = subprocess.run([sys.executable, os.path.join(tree, script), path, "--stackOutput", "--outputPath", out], env=e,
                       capture_output=True, text=True)
    dt = time.perf_counter() - t
    assert r.returncode == 0, r.stderr[-2000:]
    stages = [l[len("umx-cli-timing "):] for l in r.stderr.splitlines() if l.startswith("umx-cli-timing ")]
    print("%d x %d uint16, %s, --stackOutput: %-44s %.2f s wall (process start to exit)  %s" % (
        n, n, what, label, dt, stages[-1] if stages else ""), flush=True)
    return dt


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
    others = sys.argv[2:]
    hp, blob, mean, std = helpers.load_nuclei_dapi()
    raw = helpers.load_sample_105()[0]
    reps = (-(-n // raw.shape[0]), -(-n // raw.shape[1]))
    img = np.tile(raw, reps)[:n, :n]
    with tempfile.TemporaryDirectory() as d:
        model.save_converted(model.ModelArtefacts(hp, blob, mean, std), os.path.join(d, "models", "nucleiDAPI"))
        os.makedirs(os.path.join(d, "x", "registration"))
        path = os.path.join(d, "x", "registration", "slide.tif")
        tiffio.imsave(path, img)
        env = {"UMX_MODELS_DIR": os.path.join(d, "models")}
        for rep in range(2):
            one(ROOT, path, os.path.join(d, "out_new%d" % rep), env, "this tree, run %d" % (rep + 1), n)
            for i, tree in enumerate(others):
                one(tree, path, os.path.join(d, "out_o%d_%d" % (i, rep)), env, "%s, run %d" % (os.path.basename(tree.rstrip("/")), rep + 1), n)
        one(ROOT, path, os.path.join(d, "out_host"), dict(env, UMX_NO_RAW_PATH="1"), "this tree, host pre/post (UMX_NO_RAW_PATH=1)", n)
        # the default tool (UnMicst1-5.py, solo hyper-parameters: 116 964 tiles of 64 x 64 on this slide; the reference downloads its
        # weights at image-build time, so seeded weights stand in -- timing only) from the repo's own models/ directory
        solo_env = {"UMX_SYNTHETIC_WEIGHTS": "1"}
        for rep in range(2):
            one(ROOT, path, os.path.join(d, "out_solo%d" % rep), solo_env, "this tree, run %d" % (rep + 1), n, "UnMicst1-5.py",
                "solo hp (seeded weights)")
        # the pages of the two trees must be the same bytes
        a = os.path.join(d, "out_new0")
        for i in range(len(others)):
            b = os.path.join(d, "out_o%d_0" % i)
            for root, _, files in os.walk(a):
                for f in files:
                    pa, pb = os.path.join(root, f), os.path.join(b, os.path.relpath(os.path.join(root, f), a))
                    same = open(pa, "rb").read() == open(pb, "rb").read()
                    print("  %-40s %s" % (os.path.relpath(pa, a), "identical to the other tree's" if same else "DIFFERS"))


if __name__ == "__main__":
    main()
