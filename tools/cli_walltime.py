#!/usr/bin/env python3
"""End-to-end wall time of the reference-compatible CLI on a synthetic slide (GPU box): UnMicst.py (legacy tool, the shipped
nucleiDAPI weights from tests/golden) on an N x N uint16 TIFF, fast GPU pre/post-processing path vs the general host recipe.
usage: cli_walltime.py [N=8192]"""
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


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
    hp, blob, mean, std = helpers.load_nuclei_dapi()
    raw = helpers.load_sample_105()[0]
    reps = (-(-n // raw.shape[0]), -(-n // raw.shape[1]))
    img = np.tile(raw, reps)[:n, :n]
    with tempfile.TemporaryDirectory() as d:
        model.save_converted(model.ModelArtefacts(hp, blob, mean, std), os.path.join(d, "models", "nucleiDAPI"))
        os.makedirs(os.path.join(d, "x", "registration"))
        path = os.path.join(d, "x", "registration", "slide.tif")
        tiffio.imsave(path, img)
        for label, env in (("gpu pre/post (umx_infer_image_raw)", {}), ("host pre/post (UMX_NO_RAW_PATH=1)", {"UMX_NO_RAW_PATH": "1"})):
            e = dict(os.environ, UMX_MODELS_DIR=os.path.join(d, "models"), **env)
            t = time.perf_counter()
            r = subprocess.run([sys.executable, os.path.join(ROOT, "UnMicst.py"), path, "--stackOutput", "--outputPath",
                                os.path.join(d, "out_" + label[:3])], env=e, capture_output=True, text=True)
            dt = time.perf_counter() - t
            assert r.returncode == 0, r.stderr[-2000:]
            print("%d x %d uint16, legacy nucleiDAPI, --stackOutput: %-36s %.2f s wall (process start to exit)" % (n, n, label, dt))


if __name__ == "__main__":
    main()
