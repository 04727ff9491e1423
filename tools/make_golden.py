#!/usr/bin/env python3
"""Generate the committed golden fixtures under tests/golden/ (run ONCE, in the build container).

Needs /root/reference (read-only).  It
  1. imports the reference's own ``toolbox/PartitionOfImage.py`` (PI2D) -- with empty stub modules for the
     third-party imports that are absent here (tifffile, skimage.*) -- and records its outputs on seeded inputs;
  2. converts ``models/nucleiDAPI`` (the only shipped model with a golden output) to the canonical weight blob;
  3. packs the reference's known-answer data files ("UNet sample data/registration/105.tif" and
     "prob_maps/105_{ContoursPM,NucleiPM}_1.tif") into one compressed .npz.
Only data (inputs + expected outputs) is written; no reference source text is copied.
"""
import os
import sys
import types

import numpy as np

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
OUT = os.path.join(ROOT, "tests", "golden")
sys.path.insert(0, ROOT)
sys.dont_write_bytecode = True  # the reference tree is read-only


def import_reference_pi2d():
    for name in ("tifffile", "skimage", "skimage.io", "skimage.morphology", "skimage.transform"):
        if name not in sys.modules:
            sys.modules[name] = types.ModuleType(name)
    sys.modules["skimage.transform"].resize = lambda *a, **k: None
    sys.modules["skimage"].io = sys.modules["skimage.io"]
    import matplotlib
    matplotlib.use("Agg")
    sys.path.insert(0, REF)
    from toolbox.PartitionOfImage import PI2D  # noqa: E402
    return PI2D


def pi2d_case(PI2D, name, shape, patch, margin, mode, seed, nclass=3):
    # inputs are regenerated from the seed by tests (tests/helpers.py: pi2d_case_inputs) -- same two draws, same order
    rng = np.random.default_rng(seed)
    image = rng.random(shape)
    PI2D.setup(image, patch, margin, mode)
    T = PI2D.NumPatches
    pc = np.array(PI2D.PC, dtype=np.int32)
    patches_idx = sorted(set([0, T // 2, T - 1]))
    patches = np.stack([np.array(PI2D.getPatch(i)) for i in patches_idx])
    probs = rng.random((T, patch, patch, nclass)).astype(np.float32)
    # make a few tiles extreme to exercise fp16 rounding at both ends
    probs[0] = 1.0
    if T > 1:
        probs[1] *= 1e-4
    planes = []
    for k in range(nclass):
        PI2D.createOutput(1)
        for t in range(T):
            PI2D.patchOutput(t, probs[t, :, :, k])
        with np.errstate(divide="ignore", invalid="ignore"):
            planes.append(np.array(PI2D.getValidOutput()))
    out = np.stack(planes)
    assert out.dtype == np.float16
    np.savez_compressed(
        os.path.join(OUT, "pi2d_%s.npz" % name), shape=np.array(shape), patch=patch, margin=margin, mode=mode, pc=pc,
        W=PI2D.W, nrpi=PI2D.NRPI, ncpi=PI2D.NCPI, patches_idx=np.array(patches_idx), patches=patches,
        probs_seed=seed, nclass=nclass, stitched=out)
    print("pi2d", name, shape, patch, margin, mode, "T =", T, "padded", PI2D.NRPI, PI2D.NCPI)


def main():
    os.makedirs(OUT, exist_ok=True)
    PI2D = import_reference_pi2d()
    # sizes: non-multiples, smaller than one patch, exactly one sub-patch, multi-channel (channel-first 3-D)
    pi2d_case(PI2D, "a_300x417_p64", (300, 417), 64, 8, "accumulate", 101)
    pi2d_case(PI2D, "b_50x40_p64", (50, 40), 64, 8, "accumulate", 102)
    pi2d_case(PI2D, "c_96x96_p128", (96, 96), 128, 16, "accumulate", 103)
    pi2d_case(PI2D, "d_2x200x260_p128", (2, 200, 260), 128, 16, "accumulate", 104)
    pi2d_case(PI2D, "e_130x131_p64_replace", (130, 131), 64, 8, "replace", 105)
    pi2d_case(PI2D, "f_97x200_p64_m4", (97, 200), 64, 4, "accumulate", 106)
    pi2d_case(PI2D, "g_400x300_p256", (400, 300), 256, 32, "accumulate", 107, nclass=2)

    from unmicst_amd import model, tiffio
    art = model.load_model_dir(os.path.join(REF, "models", "nucleiDAPI"), model.GRAPH_LEGACY)
    np.savez_compressed(os.path.join(OUT, "nucleiDAPI_model.npz"), blob=art.blob, mean=art.mean, std=art.std,
                        hp=np.array([art.hp.graph, art.hp.imSize, art.hp.nChannels, art.hp.nClasses, art.hp.nOut0,
                                     art.hp.nLayers, art.hp.ks, art.hp.nExtraConvs, art.hp.featMapsFact,
                                     art.hp.downSampFact, art.hp.batchSize]))
    raw = tiffio.imread(os.path.join(REF, "UNet sample data", "registration", "105.tif"), key=0)
    cont = tiffio.imread_all(os.path.join(REF, "UNet sample data", "prob_maps", "105_ContoursPM_1.tif"))
    nuc = tiffio.imread_all(os.path.join(REF, "UNet sample data", "prob_maps", "105_NucleiPM_1.tif"))
    assert raw.dtype == np.uint16 and raw.shape == (832, 960)
    np.savez_compressed(os.path.join(OUT, "unet_sample_105.npz"), raw=raw, contours_pm=cont[0], raw_preview=cont[1],
                        nuclei_pm=nuc[0])
    print("105 fixture:", raw.shape, cont.shape, nuc.shape)


if __name__ == "__main__":
    main()
