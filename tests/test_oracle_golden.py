"""Pin the oracle (CPU restatement) against the reference's own artefacts. CPU-only."""
import numpy as np
import pytest

import helpers
from oracle import oracle, pi2d_oracle


@pytest.mark.parametrize("name", helpers.PI2D_CASES)
def test_pi2d_oracle_matches_reference_pi2d(name):
    """Bit-exact vs outputs of the imported reference toolbox/PartitionOfImage.py (tools/make_golden.py)."""
    c = helpers.load_pi2d_case(name)
    pi = pi2d_oracle.PI2DOracle(c["image"], c["patch"], c["margin"], c["mode"])
    assert (pi.nrpi, pi.ncpi) == (c["nrpi"], c["ncpi"])
    assert np.array_equal(np.array(pi.pc, dtype=np.int32), c["pc"])
    assert np.array_equal(pi.W, c["W"])
    for j, idx in enumerate(c["patches_idx"]):
        assert np.array_equal(pi.get_patch(int(idx)), c["patches"][j])
    for k in range(c["nclass"]):
        pi.create_output(1)
        for t in range(pi.num_patches):
            pi.patch_output(t, c["probs"][t, :, :, k])
        got = np.array(pi.get_valid_output())
        assert got.dtype == np.float16
        assert np.array_equal(got.view(np.uint16), c["stitched"][k].view(np.uint16)), (name, k)


def test_conv_transpose_is_adjoint_of_strided_conv():
    """orc_conv2d_transpose_s2 must be the exact transpose of the stride-2 SAME conv (TF's definition):
    <convT(x), y> == <x, conv_s2(y)> for random x, y -- checked with an independent numpy strided conv."""
    rng = np.random.default_rng(7)
    for ks in (3, 5):
        h, Cin, Cout = 6, 3, 4
        wt = rng.normal(size=(ks, ks, Cout, Cin)).astype(np.float32)  # conv2d_transpose filter [kh,kw,out,in]
        x = rng.normal(size=(1, h, h, Cin)).astype(np.float32)
        y = rng.normal(size=(1, 2 * h, 2 * h, Cout)).astype(np.float32)
        up = oracle.conv2d_transpose_s2(x, wt)
        # forward stride-2 SAME conv of y with filter wt viewed as [kh,kw,in=Cout,out=Cin]
        pad_total = max((h - 1) * 2 + ks - 2 * h, 0)
        pb = pad_total // 2
        yp = np.zeros((2 * h + pad_total, 2 * h + pad_total, Cout))
        yp[pb:pb + 2 * h, pb:pb + 2 * h] = y[0]
        f = np.zeros((h, h, Cin))
        for i in range(h):
            for j in range(h):
                win = yp[2 * i:2 * i + ks, 2 * j:2 * j + ks, :]  # [ks,ks,Cout]
                f[i, j] = np.einsum("abo,aboi->i", win, wt.astype(np.float64))
        lhs = float((up.astype(np.float64) * y).sum())
        rhs = float((x[0].astype(np.float64) * f).sum())
        assert abs(lhs - rhs) < 1e-3 * max(1.0, abs(rhs)), (ks, lhs, rhs)


def test_conv_same_matches_numpy():
    rng = np.random.default_rng(3)
    x = rng.normal(size=(2, 7, 9, 3)).astype(np.float32)
    w = rng.normal(size=(3, 3, 3, 5)).astype(np.float32)
    got = oracle.conv2d_same(x, w)
    xp = np.zeros((2, 9, 11, 3))
    xp[:, 1:-1, 1:-1] = x
    ref = np.zeros((2, 7, 9, 5))
    for a in range(3):
        for b in range(3):
            ref += np.einsum("nhwc,co->nhwo", xp[:, a:a + 7, b:b + 9], w[a, b].astype(np.float64))
    assert np.abs(got - ref).max() < 1e-5


def test_unet_oracle_matches_reference_sample_data():
    """End to end known-answer test: legacy graph + models/nucleiDAPI + PI2D on 'UNet sample data' 105.tif
    reproduces the reference's bundled prob_maps to <= 1 uint8 LSB (recipe: reference batchUnMicst.py:551-587)."""
    hp, blob, mean, std = helpers.load_nuclei_dapi()
    raw, g_cont, g_raw, g_nuc = helpers.load_sample_105()
    I = helpers.legacy_preprocess(raw)
    probs = oracle.tile_probs(hp, blob, I, mean, std, batch_size=30)
    planes = pi2d_oracle.stitch_all_classes(I.shape, hp.imSize, probs)
    for k, gold in ((1, g_cont), (2, g_nuc)):
        pm = np.uint8(255 * planes[k].astype(np.float64))
        d = np.abs(pm.astype(int) - gold.astype(int))
        assert d.max() <= 1
        assert (d == 0).mean() > 0.98
    rawI = raw.astype(np.float64) / 65535
    rawI = rawI / rawI.max()
    assert np.array_equal(np.uint8(255 * rawI), g_raw)


@pytest.mark.parametrize("name", sorted(helpers.small_hps()))
def test_oracle_forward_runs_and_is_a_distribution(name):
    from unmicst_amd import model
    hp = helpers.small_hps()[name]
    blob = model.random_blob(hp, seed=11)
    x = np.random.default_rng(5).normal(size=(2, hp.imSize, hp.imSize, hp.nChannels)).astype(np.float32)
    p = oracle.forward(hp, blob, x)
    assert p.shape == (2, hp.imSize, hp.imSize, hp.nClasses)
    assert np.allclose(p.sum(-1), 1.0, atol=1e-5) and (p >= 0).all()
    # per-sample independence: the reference relies on it when it feeds stale tiles in the last batch
    p1 = oracle.forward(hp, blob, x[1:])
    assert np.array_equal(p1[0], p[1])


@pytest.mark.parametrize("name,klass,floor", [("mousenucleiDAPI", 2, 0.45), ("CytoplasmIncell", 0, 0.55)])
def test_oracle_runs_the_other_shipped_models_on_the_sample_image(name, klass, floor):
    """The reference's two other models with weights in its tree (legacy graph, 3 x 3 kernels: the only TRAINED weights that
    take the ks = 3 transposed-convolution crop) through the C oracle on a crop of 'UNet sample data' 105.tif.  The reference
    holds no output of these models, so this pins no numbers -- it checks that the converted tensors make a model: a class
    that follows the nuclei of the image (correlation with the nucleiDAPI model's bundled nuclei map; a permuted or
    transposed tensor gives noise), probabilities that are finite and sum to one."""
    hp, blob, mean, std = helpers.load_nuclei_dapi(name)
    raw, _, _, g_nuc = helpers.load_sample_105()
    crop = helpers.legacy_preprocess(raw)[:2 * hp.imSize, :2 * hp.imSize]
    probs = oracle.tile_probs(hp, blob, crop, mean, std, batch_size=8)
    assert probs.shape[-1] == hp.nClasses and np.isfinite(probs).all()
    assert np.abs(probs.sum(-1) - 1.0).max() < 1e-5
    planes = pi2d_oracle.stitch_all_classes(crop.shape, hp.imSize, probs)
    a = planes[klass].astype(np.float64).ravel()
    assert np.corrcoef(a, g_nuc[:crop.shape[0], :crop.shape[1]].ravel().astype(np.float64))[0, 1] > floor
